// Permutohedral lattice build for one pyramid level on gfx950 (K1 + K2 of SURVEY.md §2a).
//
// Reference behaviour reproduced bit-for-bit:
//   get_keys_and_barycentric   nets/generate_data.py:56-112   -> k_point_keys
//   key2int                    nets/transforms.py:62-77       -> key2int()
//   build_it (i)  first-seen numbering   transforms.py:153-166 -> k_insert / k_flag_count /
//                                                                k_scan_sums / k_assign / k_offsets
//   build_it (ii) blur neighbours        transforms.py:168-180 -> k_neighbors
// The sequential "first seen" numbering is restated as: index(v) = rank of v among distinct
// key integers ordered by their smallest flat position 4*p+rem (atomicMin + flag prefix sum).
//
// This file is compiled with -ffp-contract=off: the float recipe must round exactly like the
// reference's MKL sgemm (FMA chain in column order).
#include "common.h"
#include <string.h>

namespace {

constexpr int TPB = 256;

__constant__ uint32_t c_elev[4][3] = {{0x3F3504F3u, 0x3ED105EBu, 0x3E93CD3Au},
                                      {0xBF3504F3u, 0x3ED105EBu, 0x3E93CD3Au},
                                      {0x00000000u, 0xBF5105EBu, 0x3E93CD3Au},
                                      {0x00000000u, 0x00000000u, 0xBF5DB3D7u}};
__constant__ int c_canon[4][4] = {{0, 1, 2, 3}, {0, 1, 2, -1}, {0, 1, -2, -1}, {0, -3, -2, -1}};
__constant__ int c_nbr[15][4] = {
    {0, 0, 0, 0},   {-1, -1, -1, 3}, {-1, -1, 3, -1}, {-2, -2, 2, 2},  {-1, 3, -1, -1},
    {-2, 2, -2, 2}, {-2, 2, 2, -2},  {-3, 1, 1, 1},   {3, -1, -1, -1}, {2, -2, -2, 2},
    {2, -2, 2, -2}, {1, -3, 1, 1},   {2, 2, -2, -2},  {1, 1, -3, 1},   {1, 1, 1, -3}};

__device__ __forceinline__ float elev(int r, int c) { return __uint_as_float(c_elev[r][c]); }

__device__ __forceinline__ int64_t key2int(const int k[4], const int *mm) {
    // mm[0..3] = mins, mm[4..7] = maxs ; int64 arithmetic, no range check (transforms.py:62-77)
    int64_t res = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        res += (int64_t)k[i] - mm[i];
        res *= ((int64_t)mm[4 + i + 1] - mm[i + 1] + 1);
    }
    res += (int64_t)k[3] - mm[3];
    return res;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

__global__ void k_init_minmax(int *mm, int nsamples, int *seg_first) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nsamples * 8) mm[i] = (i & 7) < 4 ? INT32_MAX : INT32_MIN;
    if (seg_first && i < nsamples) seg_first[i] = INT32_MAX;
}

// ---- K1: one thread per point ------------------------------------------------------------
__global__ void __launch_bounds__(TPB)
k_point_keys(const float *__restrict__ pts, int64_t cstride, int n, float scale32, float std32,
             float *__restrict__ bary, float *__restrict__ emg_out, int64_t emg_ps, int64_t emg_rs,
             int4 *__restrict__ keys, int *__restrict__ mm, const int *__restrict__ sid) {
    int p = blockIdx.x * TPB + threadIdx.x;
    int kmin[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
    int kmax[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    if (p < n) {
        float pos[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) pos[c] = __fmul_rn(pts[c * cstride + p], scale32);
        float el[4], gr[4], emg[4];
        int rank[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float acc = __fmul_rn(elev(r, 0), pos[0]);
            acc = __fmaf_rn(elev(r, 1), pos[1], acc);
            acc = __fmaf_rn(elev(r, 2), pos[2], acc);
            el[r] = __fmul_rn(acc, std32);
            gr[r] = __fmul_rn(rintf(__fmul_rn(el[r], 0.25f)), 4.0f);
            emg[r] = __fsub_rn(el[r], gr[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int c = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) c += (emg[j] > emg[r] || (emg[j] == emg[r] && j < r)) ? 1 : 0;
            rank[r] = c;
        }
        float rs = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(gr[0], gr[1]), gr[2]), gr[3]), 0.25f);
        float sign = (rs > 0.0f) ? -1.0f : ((rs < 0.0f) ? 1.0f : 0.0f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float rf = (float)rank[r];
            bool cond = ((rf >= __fsub_rn(4.0f, rs)) && (rs > 0.0f)) || ((rf < -rs) && (rs < 0.0f));
            float adj = cond ? __fmul_rn(4.0f, sign) : 0.0f;
            gr[r] = __fadd_rn(gr[r], adj);
            rank[r] += (int)adj;
            rank[r] += (int)rs;
        }
        float b5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) emg[r] = __fsub_rn(el[r], gr[r]);
        // barycentric[d0 - rank] += e ; barycentric[d1 - rank] -= e   (:99-100); rank is a
        // permutation of 0..3, so every slot gets at most one += and then one -=
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 5; ++j) if (3 - rank[r] == j) b5[j] = __fadd_rn(b5[j], emg[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 5; ++j) if (4 - rank[r] == j) b5[j] = __fsub_rn(b5[j], emg[r]);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) b5[j] = __fmul_rn(b5[j], 0.25f);
        b5[0] = __fadd_rn(b5[0], __fadd_rn(1.0f, b5[4]));
#pragma unroll
        for (int r = 0; r < 4; ++r) bary[(int64_t)r * n + p] = b5[r];
        if (emg_rs == 1 && (emg_ps & 3) == 0)       // channels 0..3 of the level's feature row: one 16-B store
            *reinterpret_cast<float4 *>(emg_out + (int64_t)p * emg_ps) = make_float4(emg[0], emg[1], emg[2], emg[3]);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) emg_out[(int64_t)p * emg_ps + r * emg_rs] = emg[r];
        }
        int g[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = (int)gr[c];
#pragma unroll
        for (int rem = 0; rem < 4; ++rem) {
            int k[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int cv = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (rank[c] == q) cv = c_canon[q][rem];
                k[c] = g[c] + cv;
                kmin[c] = min(kmin[c], k[c]);
                kmax[c] = max(kmax[c], k[c]);
            }
            keys[(int64_t)p * 4 + rem] = make_int4(k[0], k[1], k[2], k[3]);
        }
    }
    if (sid) {      // several samples in one launch: per-sample extrema
        const int b = p < n ? sid[p] : -1;
        // the usual case - the whole block belongs to one sample: LDS reduction, 8 atomics per block (the per-wave
        // form put 8 atomics per wave on the same 8 words per sample: ~65k serialised atomics at level 0, batch 4)
        const int bb = sid[(long long)blockIdx.x * TPB < n ? (long long)blockIdx.x * TPB : 0];
        if (__syncthreads_and(b == bb || b < 0)) {
            __shared__ int bmin[4][TPB / 64], bmax[4][TPB / 64];
            const int lane_ = threadIdx.x & 63, w_ = threadIdx.x >> 6;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int a = kmin[c], z = kmax[c];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); z = max(z, __shfl_xor(z, o)); }
                if (lane_ == 0) { bmin[c][w_] = a; bmax[c][w_] = z; }
            }
            __syncthreads();
            if (threadIdx.x < 4) {
                const int c = threadIdx.x;
                int a = bmin[c][0], z = bmax[c][0];
                for (int i = 1; i < TPB / 64; ++i) { a = min(a, bmin[c][i]); z = max(z, bmax[c][i]); }
                atomicMin(&mm[8 * bb + c], a);
                atomicMax(&mm[8 * bb + 4 + c], z);
            }
            return;
        }
        const int b0 = __shfl(b, 0);
        if (__all(b == b0 || b < 0)) {           // a block straddling two samples: per-wave reduction
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int a = kmin[c], z = kmax[c];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); z = max(z, __shfl_xor(z, o)); }
                if ((threadIdx.x & 63) == 0 && b0 >= 0) { atomicMin(&mm[8 * b0 + c], a); atomicMax(&mm[8 * b0 + 4 + c], z); }
            }
        } else if (b >= 0) {                     // a wave straddling two samples: per-lane atomics
#pragma unroll
            for (int c = 0; c < 4; ++c) { atomicMin(&mm[8 * b + c], kmin[c]); atomicMax(&mm[8 * b + 4 + c], kmax[c]); }
        }
        return;
    }
    // block reduce min / max, one atomic per block and coordinate
    __shared__ int smin[4][TPB / 64], smax[4][TPB / 64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int a = kmin[c], b = kmax[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); b = max(b, __shfl_xor(b, o)); }
        if (lane == 0) { smin[c][w] = a; smax[c][w] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        int c = threadIdx.x, a = smin[c][0], b = smax[c][0];
        for (int i = 1; i < TPB / 64; ++i) { a = min(a, smin[c][i]); b = max(b, smax[c][i]); }
        atomicMin(&mm[c], a);
        atomicMax(&mm[4 + c], b);
    }
}

// ---- K2a: hash insert, remember the smallest flat position per distinct key -----------------
__global__ void __launch_bounds__(TPB)
k_insert(const int4 *__restrict__ keys, int n4, const int *__restrict__ mm,
         unsigned long long *__restrict__ hkeys, int *__restrict__ minpos, int64_t hmask,
         int *__restrict__ slot, const int *__restrict__ sid, int nsamples) {
    int f = blockIdx.x * TPB + threadIdx.x;
    if (f >= n4) return;
    int4 kk = keys[f];
    int k[4] = {kk.x, kk.y, kk.z, kk.w};
    const int b = sid ? sid[f >> 2] : 0;
    // the map key is (key integer of the sample, sample): vertices of different samples never merge
    unsigned long long ki = (unsigned long long)(key2int(k, mm + 8 * b) * nsamples + b);
    uint64_t h = mix64(ki) & (uint64_t)hmask;
    const unsigned long long EMPTY = ~0ULL;
    while (true) {
        unsigned long long prev = atomicCAS(&hkeys[h], EMPTY, ki);
        if (prev == EMPTY || prev == ki) break;
        h = (h + 1) & (uint64_t)hmask;
    }
    atomicMin(&minpos[h], f);
    slot[f] = (int)h;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int *total) {
    // 256 threads, one value each -> exclusive prefix; *total = block sum
    __shared__ int wsum[TPB / 64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < TPB / 64; ++i) { if (i < w) base += wsum[i]; tot += wsum[i]; }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// ---- K2b: count first-seen positions per block of 1024 flat positions ----------------------
__global__ void __launch_bounds__(TPB)
k_flag_count(const int *__restrict__ slot, const int *__restrict__ minpos, int n4,
             int *__restrict__ bsum) {
    int base = blockIdx.x * (TPB * 4) + threadIdx.x * 4, c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { int f = base + j; if (f < n4 && minpos[slot[f]] == f) ++c; }
    int tot; block_exclusive_scan(c, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// ---- K2c: exclusive scan of the block sums (single block), writes H -------------------------
__global__ void __launch_bounds__(TPB)
k_scan_sums(int *__restrict__ bsum, int nb, int *__restrict__ H_out) {
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int s = 0; s < nb; s += TPB) {
        int i = s + threadIdx.x, v = (i < nb) ? bsum[i] : 0, tot;
        int ex = block_exclusive_scan(v, &tot);
        int carry = carry_s;
        if (i < nb) bsum[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *H_out = carry_s;
}

// ---- K2d: number the vertices, emit their keys and the next level's points ------------------
__global__ void __launch_bounds__(TPB)
k_assign(const int4 *__restrict__ keys, const int *__restrict__ slot, const int *__restrict__ minpos,
         int n4, const int *__restrict__ bsum, int *__restrict__ hvals, int4 *__restrict__ vkeys,
         float *__restrict__ pts_next, int64_t cap, float div32, const int *__restrict__ sid,
         int *__restrict__ vsid, int *__restrict__ seg_first) {
    int base = blockIdx.x * (TPB * 4) + threadIdx.x * 4, c = 0;
    bool fl[4];
    int sl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int f = base + j;
        fl[j] = false;
        if (f < n4) { sl[j] = slot[f]; fl[j] = (minpos[sl[j]] == f); }
        c += fl[j] ? 1 : 0;
    }
    int tot;
    int idx = bsum[blockIdx.x] + block_exclusive_scan(c, &tot);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!fl[j]) continue;
        int4 kk = keys[base + j];
        hvals[sl[j]] = idx;
        vkeys[idx] = kk;
        if (sid) {
            const int f = base + j, pp = f >> 2, b = sid[pp];
            vsid[idx] = b;
            // vertices are numbered sample-major, and the very first key of a sample is always new:
            // its index is the sample's first vertex index
            if ((f & 3) == 0 && (pp == 0 || sid[pp - 1] != b)) seg_first[b] = idx;
        }
        // generate_data.py:176-178: key (as fp32) / float32(std*scale), then E^T . (4-term fma chain)
        float kf[4] = {__fdiv_rn((float)kk.x, div32), __fdiv_rn((float)kk.y, div32),
                       __fdiv_rn((float)kk.z, div32), __fdiv_rn((float)kk.w, div32)};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float acc = __fmul_rn(elev(0, q), kf[0]);
            acc = __fmaf_rn(elev(1, q), kf[1], acc);
            acc = __fmaf_rn(elev(2, q), kf[2], acc);
            acc = __fmaf_rn(elev(3, q), kf[3], acc);
            pts_next[q * cap + idx] = acc;
        }
        ++idx;
    }
}

// ---- K2e: lattice_offset[rem][p] ----------------------------------------------------------------
__global__ void __launch_bounds__(TPB)
k_offsets(const int *__restrict__ slot, const int *__restrict__ hvals, int n, int *__restrict__ off) {
    int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    int4 s = reinterpret_cast<const int4 *>(slot)[p];
    off[p] = hvals[s.x];
    off[(int64_t)n + p] = hvals[s.y];
    off[(int64_t)2 * n + p] = hvals[s.z];
    off[(int64_t)3 * n + p] = hvals[s.w];
}

// ---- K2f: 15 blur neighbours per vertex, one thread per (vertex, offset) -----------------------
__global__ void __launch_bounds__(TPB)
k_neighbors(const int4 *__restrict__ vkeys, const int *__restrict__ mm,
            const unsigned long long *__restrict__ hkeys, const int *__restrict__ hvals, int64_t hmask,
            const int *__restrict__ H_dev, int *__restrict__ nbr, const int *__restrict__ vsid, int nsamples) {
    int H = *H_dev;
    for (int64_t g = (int64_t)blockIdx.x * TPB + threadIdx.x; g < (int64_t)H * 16;
         g += (int64_t)gridDim.x * TPB) {
        int h = (int)(g >> 4), t = (int)(g & 15);
        int res = -1;
        if (t < 15) {
            int4 kk = vkeys[h];
            int k[4] = {kk.x + c_nbr[t][0], kk.y + c_nbr[t][1], kk.z + c_nbr[t][2], kk.w + c_nbr[t][3]};
            const int b = vsid ? vsid[h] : 0;
            int64_t ki = key2int(k, mm + 8 * b);
            if (ki >= 0) {   // every inserted key integer is >= 0
                ki = ki * nsamples + b;
                uint64_t s = mix64((uint64_t)ki) & (uint64_t)hmask;
                while (true) {
                    unsigned long long cur = hkeys[s];
                    if (cur == ~0ULL) break;
                    if (cur == (unsigned long long)ki) { res = hvals[s]; break; }
                    s = (s + 1) & (uint64_t)hmask;
                }
            }
        }
        nbr[g] = res;
    }
}

}  // namespace

static int64_t ws_off_keys(int n) { (void)n; return 0; }
static int64_t ws_off_slot(int n) { return (int64_t)n * 64; }
static int64_t ws_off_bsum(int n) { return ws_off_slot(n) + (int64_t)n * 16; }
static int64_t ws_off_minpos(int n) { return ws_off_bsum(n) + (((int64_t)cdiv((int64_t)n * 4, 1024) + 64) * 4 + 255) / 256 * 256; }

extern "C" int64_t efgh_lattice_hash_capacity(int32_t n_in) {
    int64_t c = 1024;
    while (c < (int64_t)n_in * 8) c <<= 1;
    return c;
}

extern "C" int64_t efgh_lattice_workspace_bytes(int32_t n_in) {
    return ws_off_minpos(n_in) + efgh_lattice_hash_capacity(n_in) * 4 + 256;
}

static int lattice_build_impl(const float *pts, int64_t pts_cstride, int32_t n, float scale32,
                              float div32, float *bary, float *emg, int64_t emg_ps, int64_t emg_rs,
                              int32_t *off, int32_t *vkeys, float *pts_next, int32_t *minmax,
                              int64_t *hash_keys, int32_t *hash_vals, int64_t hcap, int32_t *H_out,
                              void *workspace, const int32_t *sid, int32_t nsamples, int32_t *vsid,
                              int32_t *seg_first, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(n > 0 && n < (1 << 28));
    EFGH_CHECK_ARG(nsamples >= 1 && (nsamples == 1 || (sid && vsid && seg_first)));
    if (nsamples == 1) sid = nullptr;
    EFGH_CHECK_ARG(hcap >= (int64_t)n * 8 && (hcap & (hcap - 1)) == 0);
    EFGH_CHECK_ARG(pts && bary && emg && off && vkeys && pts_next && minmax && hash_keys && hash_vals && H_out && workspace);
    char *ws = (char *)workspace;
    int4 *keys = (int4 *)(ws + ws_off_keys(n));
    int *slot = (int *)(ws + ws_off_slot(n));
    int *bsum = (int *)(ws + ws_off_bsum(n));
    int *minpos = (int *)(ws + ws_off_minpos(n));
    const uint32_t std_bits = 0x405105ECu;         // float32(4*sqrt(2/3)), generate_data.py:19
    float std32;
    memcpy(&std32, &std_bits, 4);
    int n4 = n * 4, nb = cdiv(n4, TPB * 4);
    hipError_t e1 = hipMemsetAsync(hash_keys, 0xFF, (size_t)hcap * 8, st);
    hipError_t e2 = hipMemsetAsync(minpos, 0x7F, (size_t)hcap * 4, st);
    if (e1 != hipSuccess || e2 != hipSuccess) {
        efgh_set_error("lattice: memset failed: %s / %s (n=%d hcap=%lld)", hipGetErrorString(e1),
                       hipGetErrorString(e2), n, (long long)hcap);
        return EFGH_E_LAUNCH;
    }
    k_init_minmax<<<cdiv(nsamples * 8, 64), 64, 0, st>>>(minmax, nsamples, sid ? seg_first : nullptr);
    k_point_keys<<<cdiv(n, TPB), TPB, 0, st>>>(pts, pts_cstride, n, scale32, std32, bary, emg, emg_ps,
                                              emg_rs, keys, minmax, sid);
    k_insert<<<cdiv(n4, TPB), TPB, 0, st>>>(keys, n4, minmax, (unsigned long long *)hash_keys, minpos,
                                           hcap - 1, slot, sid, nsamples);
    k_flag_count<<<nb, TPB, 0, st>>>(slot, minpos, n4, bsum);
    k_scan_sums<<<1, TPB, 0, st>>>(bsum, nb, H_out);
    k_assign<<<nb, TPB, 0, st>>>(keys, slot, minpos, n4, bsum, hash_vals, (int4 *)vkeys, pts_next,
                                 (int64_t)n * 4, div32, sid, vsid, seg_first);
    k_offsets<<<cdiv(n, TPB), TPB, 0, st>>>(slot, hash_vals, n, off);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_lattice_build(const float *pts, int64_t pts_cstride, int32_t n, float scale32,
                                  float div32, float *bary, float *emg, int64_t emg_ps, int64_t emg_rs,
                                  int32_t *off, int32_t *vkeys, float *pts_next, int32_t *minmax,
                                  int64_t *hash_keys, int32_t *hash_vals, int64_t hcap, int32_t *H_out,
                                  void *workspace, void *stream_) {
    return lattice_build_impl(pts, pts_cstride, n, scale32, div32, bary, emg, emg_ps, emg_rs, off, vkeys, pts_next,
                              minmax, hash_keys, hash_vals, hcap, H_out, workspace, nullptr, 1, nullptr, nullptr,
                              stream_);
}

extern "C" int efgh_lattice_build_batched(const float *pts, int64_t pts_cstride, int32_t n, float scale32,
                                          float div32, float *bary, float *emg, int64_t emg_ps, int64_t emg_rs,
                                          int32_t *off, int32_t *vkeys, float *pts_next, int32_t *minmax,
                                          int64_t *hash_keys, int32_t *hash_vals, int64_t hcap, int32_t *H_out,
                                          void *workspace, const int32_t *sid, int32_t nsamples, int32_t *vsid,
                                          int32_t *seg_first, void *stream_) {
    return lattice_build_impl(pts, pts_cstride, n, scale32, div32, bary, emg, emg_ps, emg_rs, off, vkeys, pts_next,
                              minmax, hash_keys, hash_vals, hcap, H_out, workspace, sid, nsamples, vsid, seg_first,
                              stream_);
}

extern "C" int efgh_lattice_neighbors_batched(const int32_t *vkeys, const int32_t *minmax, const int64_t *hash_keys,
                                              const int32_t *hash_vals, int64_t hcap, const int32_t *H_dev,
                                              int32_t h_bound, int32_t *nbr, const int32_t *vsid, int32_t nsamples,
                                              void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(vkeys && minmax && hash_keys && hash_vals && H_dev && nbr && h_bound > 0 && nsamples >= 1);
    EFGH_CHECK_ARG(nsamples == 1 || vsid);
    int grid = cdiv((int64_t)h_bound * 16, TPB);
    if (grid > 4096) grid = 4096;
    k_neighbors<<<grid, TPB, 0, st>>>((const int4 *)vkeys, minmax, (const unsigned long long *)hash_keys,
                                      hash_vals, hcap - 1, H_dev, nbr, nsamples > 1 ? vsid : nullptr, nsamples);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_lattice_neighbors(const int32_t *vkeys, const int32_t *minmax,
                                      const int64_t *hash_keys, const int32_t *hash_vals, int64_t hcap,
                                      const int32_t *H_dev, int32_t h_bound, int32_t *nbr, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(vkeys && minmax && hash_keys && hash_vals && H_dev && nbr && h_bound > 0);
    int grid = cdiv((int64_t)h_bound * 16, TPB);
    if (grid > 4096) grid = 4096;
    k_neighbors<<<grid, TPB, 0, st>>>((const int4 *)vkeys, minmax, (const unsigned long long *)hash_keys,
                                      hash_vals, hcap - 1, H_dev, nbr, nullptr, 1);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
