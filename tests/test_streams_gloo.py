"""World-size-2 gloo test (CPU) of the stream-parallel sharding used by bench.py:
every stream is owned by exactly one rank, seeds do not depend on the layout, and
the final gather returns rows in global stream order."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visual_odometry_amd import streams


def _worker(rank, world, port, n_streams, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owned = streams.shard_streams(n_streams, rank, world)
    # per-stream "result": derived from the stream's seed only
    local = torch.tensor([[streams.stream_seed(2021, s), s * 10] for s in owned], dtype=torch.int64).reshape(-1, 2)
    full = streams.gather_per_stream(local, owned, n_streams, world)
    t = streams.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    np.save(os.path.join(out_dir, f"t{rank}.npy"), np.array([t]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_covers_every_stream_once():
    for n, w in [(64, 8), (7, 2), (5, 8), (1, 1)]:
        owned = [streams.shard_streams(n, r, w) for r in range(w)]
        flat = sorted(s for o in owned for s in o)
        assert flat == list(range(n))
        assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


def test_gather_world2_gloo(tmp_path):
    n_streams, world, port = 7, 2, 29541
    mp.spawn(_worker, args=(world, port, n_streams, str(tmp_path)), nprocs=world, join=True)
    expect = np.array([[2021 + s, s * 10] for s in range(n_streams)])
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"r{r}.npy"), expect)
        assert np.load(tmp_path / f"t{r}.npy")[0] == 2.0
