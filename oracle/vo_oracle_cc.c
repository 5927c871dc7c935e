/*
 * CPU restatement (test infrastructure, never on the product path) of
 *   cv.connectedComponentsWithStats(dst)          reference vo_pipeline/featureExtraction.py:45
 * as far as FeatureExtractor.harris uses it: 8-connectivity, label 0 = background, one label per
 * component, per label its area and centroid (sum of coordinates / area in float64; an empty
 * background is placed at (0, 0)).
 *
 * Numbering.  order 0: components in raster order of their first pixel (what a pixel-wise two-pass
 * labeller such as scipy.ndimage.label yields -- cross-checked in tests/test_oracle_cc.py).
 * order 1: in raster order of their first 2x2 block (block rows 2r, 2r+1; block columns 2c, 2c+1).
 * That is the numbering of the block-based decision-tree labeller (Bolelli et al., "Spaghetti
 * labeling") that OpenCV >= 4.5.2 -- the reference pins opencv-python-headless 4.5.4.60,
 * environment.yml:93 -- runs for 8-connectivity: it scans 2x2 blocks in raster order, creates a new
 * provisional label exactly when a block with foreground touches no earlier labelled block, merges
 * towards the smaller provisional label, and numbers final labels by ascending smallest
 * provisional label, i.e. by each component's first block in scan order.
 * PARITY UNPINNED against cv2 itself (not installed; the reference ships no vectors).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int uf_find(int *parent, int i)
{
    while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
    return i;
}

static void uf_join(int *parent, int a, int b)
{
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a < b) parent[b] = a; else if (b < a) parent[a] = b;
}

typedef struct { int64_t key; int root; } cc_entry;
static int cc_cmp(const void *a, const void *b)
{
    const int64_t ka = ((const cc_entry *)a)->key, kb = ((const cc_entry *)b)->key;
    return ka < kb ? -1 : ka > kb;
}

/* labels (h*w, tight, may be NULL); centroids (max_labels*2 doubles, x then y) and areas (max_labels)
 * receive rows 0 .. min(n_labels, max_labels) - 1.  Returns n_labels (components + 1). */
int vo_or_connected_components(const uint8_t *mask, int h, int w, int64_t stride, int order, int32_t *labels,
                               double *centroids, int32_t *areas, int max_labels)
{
    const int n = h * w, wb = (w + 1) / 2;
    int *parent = (int *)malloc(sizeof(int) * (size_t)n);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int p = y * w + x;
            parent[p] = p;
            if (!mask[(int64_t)y * stride + x]) { parent[p] = -1; continue; }
            if (x > 0 && parent[p - 1] >= 0) uf_join(parent, p, p - 1);
            if (y > 0) {
                if (x > 0 && parent[p - w - 1] >= 0) uf_join(parent, p, p - w - 1);
                if (parent[p - w] >= 0) uf_join(parent, p, p - w);
                if (x < w - 1 && parent[p - w + 1] >= 0) uf_join(parent, p, p - w + 1);
            }
        }
    /* per root: the key of the pixel (order 0) or block (order 1) at which a sequential scan meets it first */
    int64_t *key = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int ncomp = 0;
    for (int p = 0; p < n; ++p) key[p] = -1;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int p = y * w + x;
            if (parent[p] < 0) continue;
            const int r = uf_find(parent, p);
            const int64_t k = order == 0 ? (int64_t)p : (int64_t)(y >> 1) * wb + (x >> 1);
            if (key[r] < 0) { key[r] = k; ++ncomp; }
            else if (k < key[r]) key[r] = k;
        }
    cc_entry *ent = (cc_entry *)malloc(sizeof(cc_entry) * (size_t)(ncomp + 1));
    int m = 0;
    for (int p = 0; p < n; ++p)
        if (parent[p] == p && key[p] >= 0) { ent[m].key = key[p]; ent[m].root = p; ++m; }
    qsort(ent, (size_t)m, sizeof(cc_entry), cc_cmp);       /* keys are distinct: two components never share a first block */
    int *number = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < m; ++i) number[ent[i].root] = i + 1;
    int64_t *sx = (int64_t *)calloc((size_t)m + 1, sizeof(int64_t)), *sy = (int64_t *)calloc((size_t)m + 1, sizeof(int64_t));
    int64_t *ar = (int64_t *)calloc((size_t)m + 1, sizeof(int64_t));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int p = y * w + x;
            const int l = parent[p] < 0 ? 0 : number[uf_find(parent, p)];
            if (labels) labels[p] = l;
            sx[l] += x; sy[l] += y; ar[l] += 1;
        }
    for (int l = 0; l <= m && l < max_labels; ++l) {
        const double d = (double)(ar[l] ? ar[l] : 1);
        if (centroids) { centroids[2 * l] = (double)sx[l] / d; centroids[2 * l + 1] = (double)sy[l] / d; }
        if (areas) areas[l] = (int32_t)ar[l];
    }
    free(parent); free(key); free(ent); free(number); free(sx); free(sy); free(ar);
    return m + 1;
}
