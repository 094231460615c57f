/* oracle/fps_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Sequential farthest point sampling, the contract of gr_fps (include/gaussreg_hip.h): fp32
 * ((dx*dx + dy*dy) + dz*dz) without FMA, running minimum per point, FIRST maximum on ties.
 * Same arithmetic as oracle/matching_np.py:farthest_point_sampling (which it is tested against in
 * tests/test_oracle_next.py); exists because 200 000 -> 30 000 takes a minute in NumPy and seconds here.
 * The reference calls fpsample.bucket_fps_kdline_sampling (demo.py:46), a third-party package absent
 * from /root/reference: PARITY UNPINNED -- this pins the build's own sequential definition only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

int64_t oracle_fps(const float* pts, int64_t n, int64_t k, int64_t start, int64_t* out)
{
    if (n <= 0 || k <= 0) return 0;
    float* d = (float*)malloc((size_t)n * sizeof(float));
    if (!d) return -1;
    for (int64_t i = 0; i < n; ++i) d[i] = INFINITY;
    int64_t last = start;
    out[0] = start;
    for (int64_t j = 1; j < k; ++j) {
        const float cx = pts[3 * last], cy = pts[3 * last + 1], cz = pts[3 * last + 2];
        float best = -1.0f;
        int64_t arg = 0;
        for (int64_t i = 0; i < n; ++i) {
            const float dx = pts[3 * i] - cx, dy = pts[3 * i + 1] - cy, dz = pts[3 * i + 2] - cz;
            const float dd = (dx * dx + dy * dy) + dz * dz;
            const float m = dd < d[i] ? dd : d[i];
            d[i] = m;
            if (m > best) { best = m; arg = i; }      /* strict: first maximum wins */
        }
        out[j] = arg;
        last = arg;
    }
    free(d);
    return k;
}
