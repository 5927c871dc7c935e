"""Path accuracy of the ContinuousVO loop over 60 frames of the synthetic box scene with both bootstrap routes
(SIFT + matching: the reference's; corners + KLT), several scene seeds, and what each keyframe's bootstrap had to
work with (matches, RANSAC inliers, points kept after the feasibility mask)."""
import json, logging, os, re, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import params, synth
from visual_odometry_amd.utils import loadData
from visual_odometry_amd.utils.matrix import hom_inv
from visual_odometry_amd.vo_pipeline.continuousVO import ContinuousVO

params.use_dataset("PARKING")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60


class Grab(logging.Handler):
    def __init__(self):
        super().__init__()
        self.rows = []

    def emit(self, rec):
        self.rows.append(rec.getMessage())


for seed in (5, 0, 1, 2):
    scene = synth.BoxScene(n_frames=N, seed=seed)
    frames = list(scene.frames())
    for route in ("sift", "klt"):
        grab = Grab()
        logging.getLogger().addHandler(grab)
        logging.getLogger().setLevel(logging.INFO)
        data = loadData.Dataset(np.asarray(scene.K, np.float32), iter(frames), np.asarray(scene.poses, np.float32))
        vo = ContinuousVO(data, rescale_bootstrap=True, bootstrap_correspondences=route)
        ratios = []
        while True:
            fs = vo.step()
            if fs is None:
                break
            ratios.append(fs.num_tracked_kps)
        logging.getLogger().removeHandler(grab)
        idx = list(range(vo.frames_to_skip, N))
        est = [np.asarray(vo.frame_queue.get(N - 1 - i).pose, np.float64) for i in idx]
        c_est = np.array([hom_inv(T)[:3, 3] for T in est])
        c_gt = scene.poses[idx, :3, 3]
        scale = float((c_gt * c_est).sum() / (c_est * c_est).sum())
        err = np.linalg.norm(c_est * scale - c_gt, axis=1)
        rot = [np.degrees(np.arccos(np.clip((np.trace(T[:3, :3] @ scene.poses[i, :3, :3]) - 1) / 2, -1, 1))) for T, i in zip(est, idx)]
        matches = [int(re.search(r"matches: (\d+)", m).group(1)) for m in grab.rows if "matches:" in m]
        rej = [tuple(map(int, re.search(r"(\d+) of (\d+) points rejected", m).groups())) for m in grab.rows if "points rejected" in m]
        print(json.dumps({"seed": seed, "route": route, "frames": N, "keyframes": [k.idx for k in vo.keyframes],
                          "rel_position_rms": float(np.sqrt(np.mean(err ** 2)) / np.linalg.norm(c_gt[-1])), "max_rotation_deg": float(max(rot)),
                          "matches_per_bootstrap": matches, "ransac_inliers_and_rejected_per_bootstrap": [(b, a) for a, b in rej],
                          "tracked_min_median": [int(np.min(ratios[vo.frames_to_skip + 1:])), int(np.median(ratios[vo.frames_to_skip + 1:]))]}), flush=True)
