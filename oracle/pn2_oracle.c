/*
 * pn2_oracle.c -- CPU ORACLE for the PointNet++ operator stack of HOTrack's HandTrackNet.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (hotrack_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * It restates, in plain scalar C, the *semantics* of the ten CUDA kernels of the
 * reference (network/models/pointnet_lib/src/ *.cu).  Each function cites the reference
 * lines it follows.  The reference ships no tests / golden vectors for this path and its
 * CUDA sources cannot be compiled here (THC headers, no nvcc), so:
 *
 *     PARITY UNPINNED by the reference's own tests.
 *
 * What pins it instead (tests/test_oracle_*.py, tests/golden/):
 *   - the reference's own Python fallback (network/models/pointnet_utils.py, CUDA=False
 *     branch) imported in the build container, wherever the two semantics coincide;
 *   - brute-force numpy restatements written independently of this file;
 *   - a literal simulation of the FPS shared-memory tree reduction (below) against the
 *     closed-form tie key used by the HIP kernel.
 *
 * Arithmetic convention (declared, see DESIGN.md "Distance arithmetic"):
 *   squared distance = fmaf(dz,dz, fmaf(dx,dx, dy*dy)) in fp32 -- the LLVM/NVVM contraction
 *   of the literal `dx*dx + dy*dy + dz*dz` (sampling_gpu.cu:133, ball_query_gpu.cu:33,
 *   interpolate_gpu.cu:40,108) under nvcc's default --fmad=true.  Build with
 *   -ffp-contract=off so nothing else is contracted.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PN2O_OK 0
#define PN2O_EINVAL (-1)
#define PN2O_ENOMEM (-2)

static inline float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
}

/* cuda_utils.h:10-14 -- largest power of two <= work_size, clamped to [1, 1024]. */
int pn2o_opt_n_threads(int work_size) {
    if (work_size < 1) return 1;
    int p = 1;
    while (p * 2 <= work_size && p * 2 <= 1024) p *= 2;
    return p;
}

/*
 * Furthest point sampling.  sampling_gpu.cu:94-209 (kernel), :211-253 (launcher picks
 * block_size = opt_n_threads(n)), pointnet2_utils.py:28 (temp pre-filled with 1e10).
 *
 * Literal simulation of the block: "thread" tid owns points tid, tid+bs, ... and keeps
 * (best, besti) with strict `>` starting from best=-1, besti=0 (:120-138); the shared-memory
 * tree halves the active range each level and keeps the LOWER slot on ties
 * (`v2 > v1 ? i2 : i1`, :86-91, :143-203).  idx[0] = 0 (:113-115).
 * temp may be NULL (then an internal buffer pre-filled with 1e10 is used).
 */
int pn2o_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx) {
    if (b < 0 || n < 1 || m < 0 || !xyz || (!idx && m > 0)) return PN2O_EINVAL;
    if (m == 0 || b == 0) return PN2O_OK;
    const int bs = pn2o_opt_n_threads(n);
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
    float *own_temp = NULL;
    if (!temp) {
        own_temp = (float *)malloc(sizeof(float) * (size_t)n);
        if (!own_temp) { free(dists); free(dists_i); return PN2O_ENOMEM; }
    }
    if (!dists || !dists_i) { free(dists); free(dists_i); free(own_temp); return PN2O_ENOMEM; }

    for (int bi = 0; bi < b; ++bi) {
        const float *p = xyz + (size_t)bi * n * 3;
        float *t = temp ? temp + (size_t)bi * n : own_temp;
        int *out = idx + (size_t)bi * m;
        if (!temp) for (int k = 0; k < n; ++k) t[k] = 1e10f;
        int old = 0;
        out[0] = old;
        for (int j = 1; j < m; ++j) {
            const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
            for (int tid = 0; tid < bs; ++tid) {
                int besti = 0;
                float best = -1.0f;
                for (int k = tid; k < n; k += bs) {
                    const float d = sqdist(p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2], x1, y1, z1);
                    const float d2 = fminf(d, t[k]);
                    t[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int stride = bs / 2; stride >= 1; stride /= 2) {
                for (int tid = 0; tid < stride; ++tid) {
                    const float v1 = dists[tid], v2 = dists[tid + stride];
                    const int i1 = dists_i[tid], i2 = dists_i[tid + stride];
                    dists[tid] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0];
            out[j] = old;
        }
    }
    free(dists); free(dists_i); free(own_temp);
    return PN2O_OK;
}

/*
 * Closed-form FPS: same result as the simulation above, written without the thread
 * structure.  Winner of an iteration = lexicographic max of
 *   ( d2 desc, bitrev_{log2 bs}(k mod bs) asc, k asc ).
 * Used by tests to prove that the key the HIP kernel relies on equals the tree reduction.
 */
static unsigned bitrev(unsigned v, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

int pn2o_furthest_point_sampling_keyed(int b, int n, int m, const float *xyz, int *idx) {
    if (b < 0 || n < 1 || m < 0 || !xyz || (!idx && m > 0)) return PN2O_EINVAL;
    if (m == 0 || b == 0) return PN2O_OK;
    const int bs = pn2o_opt_n_threads(n);
    int lg = 0;
    while ((1 << lg) < bs) ++lg;
    float *t = (float *)malloc(sizeof(float) * (size_t)n);
    if (!t) return PN2O_ENOMEM;
    for (int bi = 0; bi < b; ++bi) {
        const float *p = xyz + (size_t)bi * n * 3;
        int *out = idx + (size_t)bi * m;
        for (int k = 0; k < n; ++k) t[k] = 1e10f;
        int old = 0;
        out[0] = 0;
        for (int j = 1; j < m; ++j) {
            const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
            float best = -1.0f;
            uint64_t bestrank = 0;
            int besti = 0;
            for (int k = 0; k < n; ++k) {
                const float d = sqdist(p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2], x1, y1, z1);
                const float d2 = fminf(d, t[k]);
                t[k] = d2;
                const uint64_t rank = ((uint64_t)bitrev((unsigned)(k % bs), lg) << 32) | (unsigned)k;
                if (d2 > best || (d2 == best && rank < bestrank)) { best = d2; bestrank = rank; besti = k; }
            }
            old = besti;
            out[j] = old;
        }
    }
    free(t);
    return PN2O_OK;
}

/*
 * Ball query.  ball_query_gpu.cu:9-45.  radius2 = radius*radius in fp32 (:23); hit iff
 * d2 < radius2 (strict, :34); on the first hit all nsample slots are filled with it
 * (:35-39); hits are stored in scan order; stop at nsample (:42).  Rows without a hit keep
 * the caller's pre-zeroed content (pointnet2_utils.py:262) -- the oracle writes the zeros
 * itself so that callers need not pre-zero.
 */
int pn2o_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                    const float *xyz, int *idx) {
    if (b < 0 || n < 1 || m < 0 || nsample < 1 || !new_xyz || !xyz || !idx) return PN2O_EINVAL;
    const float radius2 = radius * radius;
    for (int bi = 0; bi < b; ++bi) {
        const float *p = xyz + (size_t)bi * n * 3;
        for (int s = 0; s < m; ++s) {
            const float *c = new_xyz + ((size_t)bi * m + s) * 3;
            int *row = idx + ((size_t)bi * m + s) * nsample;
            for (int l = 0; l < nsample; ++l) row[l] = 0;
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                const float d2 = sqdist(c[0], c[1], c[2], p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2]);
                if (d2 < radius2) {
                    if (cnt == 0) for (int l = 0; l < nsample; ++l) row[l] = k;
                    row[cnt] = k;
                    if (++cnt >= nsample) break;
                }
            }
        }
    }
    return PN2O_OK;
}

/*
 * kNN.  interpolate_gpu.cu:9-57.  Insertion into an ascending list of k entries with
 * strict `<` (:41) => order (d asc, index asc); best[] is double initialised to 1e40 and
 * besti[] to 0 (:30-35), so unfilled slots (m < k) come out as dist = +inf (float cast of
 * 1e40), idx = 0.  k <= 200 (:30-31).  Output is the SQUARED distance; sqrt happens in
 * Python (pointnet2_utils.py:103).
 */
int pn2o_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2,
             int *idx) {
    if (b < 0 || n < 0 || m < 0 || k < 1 || k > 200 || !unknown || !known || !dist2 || !idx)
        return PN2O_EINVAL;
    double best[200];
    int besti[200];
    for (int bi = 0; bi < b; ++bi) {
        const float *kn = known + (size_t)bi * m * 3;
        for (int q = 0; q < n; ++q) {
            const float *u = unknown + ((size_t)bi * n + q) * 3;
            for (int i = 0; i < k; ++i) { best[i] = 1e40; besti[i] = 0; }
            for (int i = 0; i < m; ++i) {
                const float d = sqdist(u[0], u[1], u[2], kn[i * 3 + 0], kn[i * 3 + 1], kn[i * 3 + 2]);
                for (int j = 0; j < k; ++j) {
                    if (d < best[j]) {
                        for (int l = k - 1; l > j; --l) { best[l] = best[l - 1]; besti[l] = besti[l - 1]; }
                        best[j] = d;
                        besti[j] = i;
                        break;
                    }
                }
            }
            float *od = dist2 + ((size_t)bi * n + q) * k;
            int *oi = idx + ((size_t)bi * n + q) * k;
            for (int i = 0; i < k; ++i) { oi[i] = besti[i]; od[i] = (float)best[i]; }
        }
    }
    return PN2O_OK;
}

/* three_nn.  interpolate_gpu.cu:81-124: cascade with strict `<` (:109-120), doubles = 1e40. */
int pn2o_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                  int *idx) {
    if (b < 0 || n < 0 || m < 0 || !unknown || !known || !dist2 || !idx) return PN2O_EINVAL;
    for (int bi = 0; bi < b; ++bi) {
        const float *kn = known + (size_t)bi * m * 3;
        for (int q = 0; q < n; ++q) {
            const float *u = unknown + ((size_t)bi * n + q) * 3;
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                const float d = sqdist(u[0], u[1], u[2], kn[k * 3 + 0], kn[k * 3 + 1], kn[k * 3 + 2]);
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *od = dist2 + ((size_t)bi * n + q) * 3;
            int *oi = idx + ((size_t)bi * n + q) * 3;
            od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
            oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
        }
    }
    return PN2O_OK;
}

/* group_points forward.  group_points_gpu.cu:47-66: out[b,c,p,s] = points[b,c,idx[b,p,s]]. */
int pn2o_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                      const int *idx, float *out) {
    if (b < 0 || c < 0 || n < 1 || npoints < 0 || nsample < 0) return PN2O_EINVAL;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *src = points + ((size_t)bi * c + ci) * n;
            float *dst = out + ((size_t)bi * c + ci) * npoints * nsample;
            const int *id = idx + (size_t)bi * npoints * nsample;
            for (int e = 0; e < npoints * nsample; ++e) dst[e] = src[id[e]];
        }
    return PN2O_OK;
}

/*
 * group_points backward.  group_points_gpu.cu:8-25: grad_points[b,c,idx] += grad_out (fp32
 * atomicAdd in hardware order).  The oracle accumulates each destination in double and adds
 * the rounded sum to the existing (pre-zeroed, pointnet2_utils.py:232) content; the HIP
 * result is compared at 1e-5 relative tolerance.
 */
int pn2o_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                           const int *idx, float *grad_points) {
    if (b < 0 || c < 0 || n < 1 || npoints < 0 || nsample < 0) return PN2O_EINVAL;
    double *acc = (double *)malloc(sizeof(double) * (size_t)n);
    if (!acc) return PN2O_ENOMEM;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *g = grad_out + ((size_t)bi * c + ci) * npoints * nsample;
            const int *id = idx + (size_t)bi * npoints * nsample;
            float *dst = grad_points + ((size_t)bi * c + ci) * n;
            memset(acc, 0, sizeof(double) * (size_t)n);
            for (int e = 0; e < npoints * nsample; ++e) acc[id[e]] += (double)g[e];
            for (int k = 0; k < n; ++k) dst[k] = (float)((double)dst[k] + acc[k]);
        }
    free(acc);
    return PN2O_OK;
}

/* gather_points forward / backward.  sampling_gpu.cu:8-24, :46-63 (= group with nsample 1). */
int pn2o_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                       float *out) {
    return pn2o_group_points(b, c, n, npoints, 1, points, idx, out);
}
int pn2o_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                            const int *idx, float *grad_points) {
    return pn2o_group_points_grad(b, c, n, npoints, 1, grad_out, idx, grad_points);
}

/*
 * three_interpolate forward.  interpolate_gpu.cu:149-169:
 *   out[b,c,j] = w0*p[i0] + w1*p[i1] + w2*p[i2]   (:168)
 * evaluated as fmaf(w2,p2, fmaf(w0,p0, w1*p1)) (same contraction rule as the distance).
 * Compared at 1e-5, so the order only matters at the ulp level.
 */
int pn2o_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                           const float *weight, float *out) {
    if (b < 0 || c < 0 || m < 1 || n < 0) return PN2O_EINVAL;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *src = points + ((size_t)bi * c + ci) * m;
            float *dst = out + ((size_t)bi * c + ci) * n;
            for (int j = 0; j < n; ++j) {
                const int *id = idx + ((size_t)bi * n + j) * 3;
                const float *w = weight + ((size_t)bi * n + j) * 3;
                dst[j] = fmaf(w[2], src[id[2]], fmaf(w[0], src[id[0]], w[1] * src[id[1]]));
            }
        }
    return PN2O_OK;
}

/* three_interpolate backward.  interpolate_gpu.cu:192-214 (atomicAdd x3, :211-213). */
int pn2o_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                const float *weight, float *grad_points) {
    if (b < 0 || c < 0 || m < 1 || n < 0) return PN2O_EINVAL;
    double *acc = (double *)malloc(sizeof(double) * (size_t)m);
    if (!acc) return PN2O_ENOMEM;
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *g = grad_out + ((size_t)bi * c + ci) * n;
            float *dst = grad_points + ((size_t)bi * c + ci) * m;
            memset(acc, 0, sizeof(double) * (size_t)m);
            for (int j = 0; j < n; ++j) {
                const int *id = idx + ((size_t)bi * n + j) * 3;
                const float *w = weight + ((size_t)bi * n + j) * 3;
                for (int t = 0; t < 3; ++t) acc[id[t]] += (double)(g[j] * w[t]);
            }
            for (int k = 0; k < m; ++k) dst[k] = (float)((double)dst[k] + acc[k]);
        }
    free(acc);
    return PN2O_OK;
}
