// Permutohedral lattice build for one pyramid level, all samples of a batch, on gfx950 (K1 + K2 of SURVEY.md §2a).
//
// Reference behaviour reproduced bit-for-bit:
//   get_keys_and_barycentric   nets/generate_data.py:56-112   -> point_keys()  (k_point_keys, k_insert, k_assign)
//   key2int                    nets/transforms.py:62-77       -> key2int()
//   build_it (i)  first-seen numbering   transforms.py:153-166 -> k_insert / k_seg_* / k_place / k_sortmin /
//                                                                k_flag_count / k_scan_sums / k_assign / k_offsets
//   build_it (ii) blur neighbours        transforms.py:168-180 -> k_neighbors
// The sequential "first seen" numbering is restated as: index(v) = rank of v among the distinct key integers ordered by
// their smallest flat position f = 4*p + rem.
//
// What is different from a line-by-line GPU version (round 1), and why (tools/probes/probe_atomics.hip, profiles/r02_probes.txt):
// every global atomic on gfx950 executes at the memory side at ~26 G operations/s chip-wide whatever its scope, width or
// table size, so the cost of this build is its number of atomics per (point, remainder) entry.  Round 1 paid four (hash CAS,
// atomicMin of the position, and a count + a fill atomic to invert `off` for the splat); this version pays ONE:
//   * the key is looked up with an ordinary load first (a key never changes once written, so a match is final) and claimed
//     by CAS only when the slot is still empty (one CAS per vertex, not per entry);
//   * the one atomic per entry is `rank = atomicAdd(cnt[slot], 1)`.  It yields the entry's place in its vertex's list, so
//     the inverse of `off` (vertex -> its entries: what the splat gather walks) falls out of the build: one list segment
//     per occupied slot (exclusive scan of cnt, which also compacts the occupied slots), entries placed with plain stores,
//     every list sorted by a half-wave (fixed summation order: the forward stays bit-reproducible); its head IS the
//     smallest position;
//   * same-ADDRESS atomics serialise at ~20-70 ns each (a per-wave cursor bump measured 1.3 ms for 110 k waves), so nothing
//     funnels through one counter: key extrema leave k_point_keys as per-wave records, segments come from a scan;
//   * keys are recomputed from the point where they are needed instead of stored (64 B per point never written).
// Sizes that depend on the data (n of levels >= 1, H) are read from DEVICE memory, launch sizes come from capacities, so
// a whole pyramid can be enqueued without a host read-back between levels (efgh_amd/lattice.py).
//
// This file is compiled with -ffp-contract=off: the float recipe must round exactly like the
// reference's MKL sgemm (FMA chain in column order).
#include "common.h"
#include <string.h>

namespace {

constexpr int TPB = 256;
constexpr unsigned long long EMPTY = ~0ULL;

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

__device__ __forceinline__ int n_of(const int *n_dev, int n_cap) {
    if (!n_dev) return n_cap;
    int n = *n_dev;
    return n < n_cap ? n : n_cap;
}

__device__ __forceinline__ int sample_of(const int *sid, int pps, int p) { return sid ? sid[p] : p / pps; }

// generate_data.py:56-112 for one point: greedy lattice point g, rank, el_minus_gr, barycentric weights
struct PointKeys { float bary[4], emg[4]; int g[4], rank[4]; };

__device__ __forceinline__ void point_keys(float x, float y, float z, float scale32, float std32, PointKeys &o) {
    float pos[3] = {__fmul_rn(x, scale32), __fmul_rn(y, scale32), __fmul_rn(z, scale32)};
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
    for (int r = 0; r < 4; ++r) { o.bary[r] = b5[r]; o.emg[r] = emg[r]; o.g[r] = (int)gr[r]; o.rank[r] = rank[r]; }
}

// key of simplex vertex `rem` (generate_data.py:106): greedy + canonical[rank][rem]
__device__ __forceinline__ void entry_key(const PointKeys &pk, int rem, int k[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int cv = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (pk.rank[c] == q) cv = c_canon[q][rem];
        k[c] = pk.g[c] + cv;
    }
}

// ---- K0: clear the tables of this level ------------------------------------------------------------
__global__ void __launch_bounds__(TPB)
k_level_init(unsigned long long *__restrict__ hkeys, int *__restrict__ cnt, int64_t hcap, int *__restrict__ flags,
             int n_cap, int *__restrict__ mm, int nsamples, int *__restrict__ info, int ninfo) {
    const int64_t i0 = (int64_t)blockIdx.x * TPB + threadIdx.x, stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = i0; i < hcap; i += stride) { hkeys[i] = EMPTY; cnt[i] = 0; }
    for (int64_t i = i0; i < n_cap; i += stride) flags[i] = 0;
    if (i0 < nsamples * 8) mm[i0] = (i0 & 7) < 4 ? INT32_MAX : INT32_MIN;
    if (i0 < ninfo) info[i0] = 0;
}

// per-sample extrema of a wave whose lanes belong to more than one sample: one shuffle reduction per distinct sample, then
// eight atomics by one lane (not eight per lane: same-address atomics serialise)
__device__ __forceinline__ void wave_minmax_by_sample(int b, const int kmin[4], const int kmax[4], int *__restrict__ mm) {
    unsigned long long todo = __ballot(b >= 0);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const int bs = __shfl(b, l);
        const bool mine = b == bs;
        int lo[4], hi[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lo[c] = mine ? kmin[c] : INT32_MAX;
            hi[c] = mine ? kmax[c] : INT32_MIN;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo[c] = min(lo[c], __shfl_xor(lo[c], o)); hi[c] = max(hi[c], __shfl_xor(hi[c], o)); }
        }
        if (lane == l) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { atomicMin(&mm[8 * bs + c], lo[c]); atomicMax(&mm[8 * bs + 4 + c], hi[c]); }
        }
        todo &= ~__ballot(mine);
    }
}

// ---- K1: one thread per point: barycentric weights, el_minus_gr, per-sample key extrema.  Same-address atomics serialise
// at the memory side (~20-70 ns each), so the extrema leave the kernel as one plain 48-byte record per WAVE (part[wave] =
// sample, mins, maxs) and k_minmax_finalize folds the records; only a wave that straddles two samples (at most one per sample
// boundary) falls back to atomics.
__global__ void __launch_bounds__(TPB)
k_point_keys(const float *__restrict__ pts, int64_t cstride, const int *__restrict__ n_dev, int n_cap, float scale32,
             float std32, float4 *__restrict__ bary, float4 *__restrict__ emg, int *__restrict__ mm,
             int *__restrict__ part, const int *__restrict__ sid, int pps) {
    const int n = n_of(n_dev, n_cap);
    const int p = blockIdx.x * TPB + threadIdx.x;
    const int wave = p >> 6, lane = threadIdx.x & 63;
    if ((p & ~63) >= n) {                       // wave past the end: an empty record
        if (lane == 0) part[12 * wave] = -1;
        return;
    }
    int kmin[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
    int kmax[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    int b = -1;
    if (p < n) {
        PointKeys pk;
        point_keys(pts[p], pts[cstride + p], pts[2 * cstride + p], scale32, std32, pk);
        bary[p] = make_float4(pk.bary[0], pk.bary[1], pk.bary[2], pk.bary[3]);
        emg[p] = make_float4(pk.emg[0], pk.emg[1], pk.emg[2], pk.emg[3]);
#pragma unroll
        for (int rem = 0; rem < 4; ++rem) {
            int k[4];
            entry_key(pk, rem, k);
#pragma unroll
            for (int c = 0; c < 4; ++c) { kmin[c] = min(kmin[c], k[c]); kmax[c] = max(kmax[c], k[c]); }
        }
        b = sample_of(sid, pps, p);
    }
    const int b0 = __shfl(b, 0);                // lane 0 is a valid point here
    if (__all(b == b0 || b < 0)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                kmin[c] = min(kmin[c], __shfl_xor(kmin[c], o));
                kmax[c] = max(kmax[c], __shfl_xor(kmax[c], o));
            }
        }
        if (lane == 0) {
            int *q = part + 12 * wave;
            q[0] = b0;
#pragma unroll
            for (int c = 0; c < 4; ++c) { q[1 + c] = kmin[c]; q[5 + c] = kmax[c]; }
        }
    } else {
        if (lane == 0) part[12 * wave] = -1;
        wave_minmax_by_sample(b, kmin, kmax, mm);
    }
}

// fold the per-wave records: one thread per record, a shuffle reduction per wave of records and distinct sample (records are
// sample-major, so almost every wave holds one sample), eight atomics per wave
__global__ void __launch_bounds__(TPB)
k_minmax_finalize(const int *__restrict__ part, int nwaves, int *__restrict__ mm) {
    const int w = blockIdx.x * TPB + threadIdx.x;
    int b = -1;
    int kmin[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
    int kmax[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    if (w < nwaves) {
        const int4 *q = reinterpret_cast<const int4 *>(part + 12 * w);
        const int4 q0 = q[0], q1 = q[1], q2 = q[2];
        b = q0.x;
        kmin[0] = q0.y; kmin[1] = q0.z; kmin[2] = q0.w; kmin[3] = q1.x;
        kmax[0] = q1.y; kmax[1] = q1.z; kmax[2] = q1.w; kmax[3] = q2.x;
    }
    wave_minmax_by_sample(b, kmin, kmax, mm);
}

// ---- K2a: hash insert.  One thread per point, its four entries in flight together.  Per entry: a load of the home slot
// (a hit for all but the first entries of a vertex), CAS only on an empty slot, then the ONE atomic of the build,
// rank = atomicAdd(cnt[slot], 1).
__device__ __forceinline__ unsigned long long load_slot(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // sc1: served by the memory side, as the CAS is
}

__device__ __forceinline__ int claim_slot(unsigned long long *__restrict__ hkeys, uint64_t hmask, unsigned long long ki,
                                          uint64_t h, unsigned long long cur, int *__restrict__ info) {
    for (uint64_t probe = 0; probe <= hmask; ++probe) {
        if (cur == ki) return (int)h;
        if (cur == EMPTY) {
            const unsigned long long prev = atomicCAS(&hkeys[h], EMPTY, ki);
            if (prev == EMPTY || prev == ki) return (int)h;
        }
        h = (h + 1) & hmask;
        cur = load_slot(&hkeys[h]);
    }
    // every slot holds another key: the caller sized the table from an estimate that was too small (hash_slots).  Flag it; the
    // entry is counted on an arbitrary slot so that all sizes stay in bounds, and the host rebuilds with the default table.
    atomicOr(&info[EFGH_LATTICE_INFO_ERR], 4);
    return (int)h;
}

__global__ void __launch_bounds__(TPB)
k_insert(const float *__restrict__ pts, int64_t cstride, const int *__restrict__ n_dev, int n_cap, float scale32,
         float std32, const int *__restrict__ mm, unsigned long long *__restrict__ hkeys, int *__restrict__ cnt,
         int64_t hmask, int4 *__restrict__ slot, int4 *__restrict__ rnk, const int *__restrict__ sid, int pps,
         int nsamples, int *__restrict__ info) {
    const int n = n_of(n_dev, n_cap);
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    PointKeys pk;
    point_keys(pts[p], pts[cstride + p], pts[2 * cstride + p], scale32, std32, pk);
    const int b = sample_of(sid, pps, p);
    unsigned long long ki[4], cur[4];
    uint64_t h[4];
    int s[4], r[4];
#pragma unroll
    for (int rem = 0; rem < 4; ++rem) {
        int k[4];
        entry_key(pk, rem, k);
        // the map key is (key integer of the sample, sample): vertices of different samples never merge
        ki[rem] = (unsigned long long)(key2int(k, mm + 8 * b) * nsamples + b);
        h[rem] = mix64(ki[rem]) & (uint64_t)hmask;
    }
    // first look with an ordinary (cacheable) load: a key never changes once written, so a match is final; anything else
    // (empty, another key, or a stale cached line) goes through the memory-side load / CAS of claim_slot
#pragma unroll
    for (int rem = 0; rem < 4; ++rem) cur[rem] = hkeys[h[rem]];
#pragma unroll
    for (int rem = 0; rem < 4; ++rem)
        s[rem] = cur[rem] == ki[rem] ? (int)h[rem] : claim_slot(hkeys, (uint64_t)hmask, ki[rem], h[rem], load_slot(&hkeys[h[rem]]), info);
#pragma unroll
    for (int rem = 0; rem < 4; ++rem) r[rem] = atomicAdd(&cnt[s[rem]], 1);
    slot[p] = make_int4(s[0], s[1], s[2], s[3]);
    rnk[p] = make_int4(r[0], r[1], r[2], r[3]);
}

// ---- K2b: one list segment per occupied slot: exclusive scan of cnt over the slots (block sums, one-block scan, assign) which
// also compacts the occupied slots into `occ` (their number is the number of vertices).
constexpr int SEG_Q = 4;            // int4 per thread in the slot scan: 4096 slots per block

__global__ void __launch_bounds__(TPB)
k_seg_count(const int4 *__restrict__ cnt4, int2 *__restrict__ bsum2) {
    int sum = 0, occ = 0;
#pragma unroll
    for (int q = 0; q < SEG_Q; ++q) {
        const int4 c = cnt4[((int64_t)blockIdx.x * SEG_Q + q) * TPB + threadIdx.x];
        sum += c.x + c.y + c.z + c.w;
        occ += (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); occ += __shfl_xor(occ, o); }
    __shared__ int2 ws_[TPB / 64];
    if ((threadIdx.x & 63) == 0) ws_[threadIdx.x >> 6] = make_int2(sum, occ);
    __syncthreads();
    if (threadIdx.x == 0) {
        int2 t = ws_[0];
        for (int i = 1; i < TPB / 64; ++i) { t.x += ws_[i].x; t.y += ws_[i].y; }
        bsum2[blockIdx.x] = t;
    }
}

__device__ __forceinline__ int2 block_exclusive_scan2(int2 v, int2 *total) {
    __shared__ int2 wsum2[TPB / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int2 x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y0 = __shfl_up(x.x, o), y1 = __shfl_up(x.y, o);
        if (lane >= o) { x.x += y0; x.y += y1; }
    }
    if (lane == 63) wsum2[w] = x;
    __syncthreads();
    int2 base = make_int2(0, 0), tot = make_int2(0, 0);
#pragma unroll
    for (int i = 0; i < TPB / 64; ++i) {
        if (i < w) { base.x += wsum2[i].x; base.y += wsum2[i].y; }
        tot.x += wsum2[i].x; tot.y += wsum2[i].y;
    }
    __syncthreads();
    *total = tot;
    return make_int2(base.x + x.x - v.x, base.y + x.y - v.y);
}

__global__ void __launch_bounds__(TPB)
k_seg_scan(int2 *__restrict__ bsum2, int nb, int *__restrict__ n_occ) {
    __shared__ int2 carry_s;
    if (threadIdx.x == 0) carry_s = make_int2(0, 0);
    __syncthreads();
    for (int s = 0; s < nb; s += TPB) {
        const int i = s + threadIdx.x;
        const int2 v = i < nb ? bsum2[i] : make_int2(0, 0);
        int2 tot;
        const int2 ex = block_exclusive_scan2(v, &tot);
        const int2 carry = carry_s;
        if (i < nb) bsum2[i] = make_int2(carry.x + ex.x, carry.y + ex.y);
        __syncthreads();
        if (threadIdx.x == 0) carry_s = make_int2(carry.x + tot.x, carry.y + tot.y);
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_occ = carry_s.y;
}

// occ[k] = (slot, start, length) of the k-th occupied slot
__global__ void __launch_bounds__(TPB)
k_seg_assign(const int4 *__restrict__ cnt4, const int2 *__restrict__ bsum2, int4 *__restrict__ sstart4, int4 *__restrict__ occ) {
    int2 run = bsum2[blockIdx.x];
#pragma unroll
    for (int q = 0; q < SEG_Q; ++q) {
        const int64_t i = ((int64_t)blockIdx.x * SEG_Q + q) * TPB + threadIdx.x;
        const int4 c = cnt4[i];
        const int2 mine = make_int2(c.x + c.y + c.z + c.w, (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0));
        int2 tot;
        const int2 ex = block_exclusive_scan2(mine, &tot);
        int st = run.x + ex.x, k = run.y + ex.y;
        run.x += tot.x; run.y += tot.y;
        const int s0 = (int)(i * 4);
        int4 o;
        o.x = st; if (c.x > 0) occ[k++] = make_int4(s0, st, c.x, 0);
        st += c.x;
        o.y = st; if (c.y > 0) occ[k++] = make_int4(s0 + 1, st, c.y, 0);
        st += c.y;
        o.z = st; if (c.z > 0) occ[k++] = make_int4(s0 + 2, st, c.z, 0);
        st += c.z;
        o.w = st; if (c.w > 0) occ[k++] = make_int4(s0 + 3, st, c.w, 0);
        sstart4[i] = o;
    }
}

// ---- K2c: entries into their vertex's segment, in arrival order ------------------------------------------------
__global__ void __launch_bounds__(TPB)
k_place(const int4 *__restrict__ slot, const int4 *__restrict__ rnk, const int *__restrict__ n_dev, int n_cap,
        const int *__restrict__ sstart, int *__restrict__ list0) {
    const int n = n_of(n_dev, n_cap);
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    const int4 s = slot[p], r = rnk[p];
    list0[sstart[s.x] + r.x] = 4 * p;
    list0[sstart[s.y] + r.y] = 4 * p + 1;
    list0[sstart[s.z] + r.z] = 4 * p + 2;
    list0[sstart[s.w] + r.w] = 4 * p + 3;
}

// ---- K2d: every list sorted by flat position (rank = number of smaller ids, ids are distinct): a 32-lane half-wave per
// occupied slot.  The head of the sorted list is the position where the vertex is first seen: flag it.
__global__ void __launch_bounds__(TPB)
k_sortmin(const int4 *__restrict__ occ, const int *__restrict__ n_occ, const int *__restrict__ list0,
          int *__restrict__ list, unsigned char *__restrict__ flags) {
    const int nocc = *n_occ;
    const int lane = threadIdx.x & 63, sub = lane & 31, half = lane & 32;
    for (int64_t v0 = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 5; (v0 & ~1LL) < nocc; v0 += ((int64_t)gridDim.x * TPB) >> 5) {
        const bool have = v0 < nocc;
        const int4 rec = have ? occ[v0] : make_int4(0, 0, 0, 0);
        const int len = rec.z, s0 = rec.y;
        const int lmax = max(len, __shfl_xor(len, 32));          // both halves run the same number of shuffle rounds
        if (lmax <= 64) {                          // up to two ids per lane, ranks by shuffles only
            const int id0 = sub < len ? list0[s0 + sub] : INT32_MAX;
            const int id1 = sub + 32 < len ? list0[s0 + 32 + sub] : INT32_MAX;
            int r0 = 0, r1 = 0;
            const int n0 = min(lmax, 32);
            for (int q = 0; q < n0; ++q) {
                const int x = __shfl(id0, half + q);
                r0 += x < id0 ? 1 : 0;
                r1 += x < id1 ? 1 : 0;
            }
            for (int q = 0; q < lmax - 32; ++q) {
                const int x = __shfl(id1, half + q);
                r0 += x < id0 ? 1 : 0;
                r1 += x < id1 ? 1 : 0;
            }
            if (sub < len) {
                list[s0 + r0] = id0;
                if (r0 == 0) flags[id0] = 1;
            }
            if (sub + 32 < len) {
                list[s0 + r1] = id1;
                if (r1 == 0) flags[id1] = 1;
            }
        } else {                                   // long lists (many points in one cell): O(len^2 / 32)
            for (int e = sub; e < len; e += 32) {
                const int id = list0[s0 + e];
                int rank = 0;
                for (int q = 0; q < len; ++q) rank += list0[s0 + q] < id ? 1 : 0;
                list[s0 + rank] = id;
                if (rank == 0) flags[id] = 1;
            }
        }
    }
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

__device__ __forceinline__ int flag_count(unsigned w) { return __popc(w & 0x01010101u); }

// ---- K2e: count first-seen positions per block of 256 points (1024 flat positions) -------------------------
__global__ void __launch_bounds__(TPB)
k_flag_count(const unsigned *__restrict__ flags, const int *__restrict__ n_dev, int n_cap, int *__restrict__ bsum) {
    const int n = n_of(n_dev, n_cap);
    const int p = blockIdx.x * TPB + threadIdx.x;
    int tot;
    block_exclusive_scan(p < n ? flag_count(flags[p]) : 0, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// ---- K2f: exclusive scan of the block sums (single block), writes H -------------------------
__global__ void __launch_bounds__(TPB)
k_scan_sums(int *__restrict__ bsum, int nb, int *__restrict__ info, int h_cap) {
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
    if (threadIdx.x == 0) {
        info[EFGH_LATTICE_INFO_H] = carry_s;
        if (carry_s > h_cap) atomicOr(&info[EFGH_LATTICE_INFO_ERR], 1);
    }
}

// ---- K2g: number the vertices, emit their keys, list segments and the next level's points ------------------
__global__ void __launch_bounds__(TPB)
k_assign(const float *__restrict__ pts, int64_t cstride, const int *__restrict__ n_dev, int n_cap, float scale32,
         float std32, const unsigned *__restrict__ flags, const int4 *__restrict__ slot, const int *__restrict__ bsum,
         const int *__restrict__ cnt, const int *__restrict__ sstart, int *__restrict__ hvals, int4 *__restrict__ vkeys,
         int2 *__restrict__ vseg, float *__restrict__ pts_next, int h_cap, float div32, const int *__restrict__ sid,
         int pps, int *__restrict__ vsid, int *__restrict__ info) {
    const int n = n_of(n_dev, n_cap);
    const int p = blockIdx.x * TPB + threadIdx.x;
    const unsigned fw = p < n ? (flags[p] & 0x01010101u) : 0u;
    int tot;
    int idx = bsum[blockIdx.x] + block_exclusive_scan(__popc(fw), &tot);
    if (!fw) return;
    PointKeys pk;
    point_keys(pts[p], pts[cstride + p], pts[2 * cstride + p], scale32, std32, pk);
    const int b = sample_of(sid, pps, p);
    const int4 s4 = slot[p];
    const int sl[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
    for (int rem = 0; rem < 4; ++rem) {
        if (!(fw >> (8 * rem) & 1u)) continue;
        const int h = sl[rem];
        hvals[h] = idx;
        if (idx < h_cap) {
            int k[4];
            entry_key(pk, rem, k);
            vkeys[idx] = make_int4(k[0], k[1], k[2], k[3]);
            vseg[idx] = make_int2(sstart[h], cnt[h]);
            vsid[idx] = b;
            // vertices are numbered sample-major, and the very first key of a sample is always new:
            // its index is the sample's first vertex index
            if (rem == 0 && (p == 0 || sample_of(sid, pps, p - 1) != b)) info[EFGH_LATTICE_INFO_SEG + b] = idx;
            // generate_data.py:176-178: key (as fp32) / float32(std*scale), then E^T . (4-term fma chain)
            float kf[4] = {__fdiv_rn((float)k[0], div32), __fdiv_rn((float)k[1], div32),
                           __fdiv_rn((float)k[2], div32), __fdiv_rn((float)k[3], div32)};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float acc = __fmul_rn(elev(0, q), kf[0]);
                acc = __fmaf_rn(elev(1, q), kf[1], acc);
                acc = __fmaf_rn(elev(2, q), kf[2], acc);
                acc = __fmaf_rn(elev(3, q), kf[3], acc);
                pts_next[(int64_t)q * h_cap + idx] = acc;
            }
        }
        ++idx;
    }
}

// ---- K2h: lattice_offset[p][rem] ----------------------------------------------------------------
__global__ void __launch_bounds__(TPB)
k_offsets(const int4 *__restrict__ slot, const int *__restrict__ hvals, const int *__restrict__ n_dev, int n_cap,
          int4 *__restrict__ off) {
    const int n = n_of(n_dev, n_cap);
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= n) return;
    const int4 s = slot[p];
    off[p] = make_int4(hvals[s.x], hvals[s.y], hvals[s.z], hvals[s.w]);
}

// ---- K2i: 15 blur neighbours per vertex, one thread per (vertex, offset).  key2int has no range check
// (transforms.py:173-180): a neighbour key outside the sample's [mins, maxs] box in coordinates 1..3 aliases to the integer
// of another lattice point and may "find" that vertex.  Such hits are kept (reference behaviour) and marked: bit t of column
// 15 of the row, and an (entry, target) record in `alist` - everything else in the table is an exact, hence symmetric,
// neighbour relation, which is what lets the adjoint of the blur gather run as a gather (bcl.hip).
__global__ void __launch_bounds__(TPB)
k_neighbors(const int4 *__restrict__ vkeys, const int *__restrict__ mm, const unsigned long long *__restrict__ hkeys,
            const int *__restrict__ hvals, int64_t hmask, int *__restrict__ info, int h_cap, int *__restrict__ nbr,
            const int *__restrict__ vsid, int nsamples, int2 *__restrict__ alist, int alias_cap) {
    int H = info[EFGH_LATTICE_INFO_H];
    if (H > h_cap) H = h_cap;
    const int lane = threadIdx.x & 63;
    for (int64_t g0 = (int64_t)blockIdx.x * TPB; g0 < (int64_t)H * 16; g0 += (int64_t)gridDim.x * TPB) {
        const int64_t g = g0 + threadIdx.x;
        const int h = (int)(g >> 4), t = (int)(g & 15);
        int res = -1;
        bool aliased = false;
        if (h < H && t < 15) {
            int4 kk = vkeys[h];
            int k[4] = {kk.x + c_nbr[t][0], kk.y + c_nbr[t][1], kk.z + c_nbr[t][2], kk.w + c_nbr[t][3]};
            const int b = vsid[h];
            const int *m8 = mm + 8 * b;
            int64_t ki = key2int(k, m8);
            if (ki >= 0) {   // every inserted key integer is >= 0
                ki = ki * nsamples + b;
                uint64_t s = mix64((uint64_t)ki) & (uint64_t)hmask;
                for (int64_t probe = 0; probe <= hmask; ++probe) {
                    unsigned long long cur = hkeys[s];
                    if (cur == EMPTY) break;
                    if (cur == (unsigned long long)ki) { res = hvals[s]; break; }
                    s = (s + 1) & (uint64_t)hmask;
                }
            }
            aliased = res >= 0 && (k[1] < m8[1] || k[1] > m8[5] || k[2] < m8[2] || k[2] > m8[6] || k[3] < m8[3] || k[3] > m8[7]);
        }
        const unsigned long long am = __ballot(aliased);
        if (h < H) {
            if (t < 15) nbr[g] = res;
            else nbr[g] = (int)((am >> (lane & 48)) & 0x7FFFu);
        }
        if (am) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&info[EFGH_LATTICE_INFO_ALIAS], __popcll(am));
            base = __shfl(base, 0);
            if (aliased) {
                const int k_ = base + __popcll(am & ((1ULL << lane) - 1ULL));
                if (k_ < alias_cap) alist[k_] = make_int2((int)g, res);
                else atomicOr(&info[EFGH_LATTICE_INFO_ERR], 2);
            }
        }
    }
}

// =====================================================================================================================
// Partitioned build (round 3): the same lattice, built WITHOUT a global hash insert and without scattered global stores.
//
// The hash build above is bound by memory-side atomics (~2 per (point, corner) entry at ~26 G/s chip-wide).  Measured on this
// part while rebuilding it: 4.2 M random 4-byte STORES cost 80-100 us, 4.2 M random 4-byte LOADS 14 us - so every permutation
// below is a gather.
//   k_lat_keys     barycentric weights + one key-extrema record per block; k_lat_minmax folds the records (one workgroup).
//   k_lat_scatter  a tile of points computes its 4 entries per point, counts them per bucket in LDS (bucket = top bits of
//                  mix64(key integer): mix64 is a bijection, so equal keys meet in one bucket and the buckets are balanced whatever
//                  the key distribution), and writes them bucket-sorted into the TILE's own window of `ent` (a local scatter that
//                  the L2 merges into full lines) + the tile's row of bucket offsets + where[f] = (bucket, rank in the tile's run).
//   k_lat_bucket   one workgroup per bucket gathers its runs from all tiles (<= maxe entries) and groups them by key entirely in
//                  LDS (open-addressing table + counts + segment scan + place + rank-sort of every vertex's entries, LDS atomics
//                  only), which leaves the vertex lists (`list`, ascending flat position per vertex: the splat's fixed summation
//                  order), one bit per flat position marking the first-seen entry of every vertex, a read-only image of the
//                  bucket's table for the neighbour probes, and the bucket-local vertex of every arrival position.
//   k_lat_rank     first-seen numbering = prefix count over those bits.
//   k_lat_number   vertex numbers (to the tables) and their inverse.
//   k_lat_nbr      15 neighbour probes per vertex; the same launch gathers lattice_offset per point through where[] and emits
//                  the vertex records (list segment, sample, next level's point) in number order.
// A bucket that overflows maxe entries or its table sets bit 2 of info[ERR]: the caller rebuilds with the hash build.
constexpr int SMAX = 2048;          // slots of a bucket's table
constexpr int NBMAX = 8192;         // buckets (LDS counters of k_lat_scatter)
constexpr int NTMAX = 1024;         // tiles (LDS run table of k_lat_bucket)
constexpr int STP = 512;            // threads of k_lat_scatter
constexpr int MAXE_BIG = 8192;      // entries of a bucket k_lat_bucket_big can hold (152 KB of LDS, 1024 threads)
constexpr int BIGCAP = 1024;        // such buckets per level

struct LatPart {                    // workspace of the partitioned build
    unsigned long long *ent;        // [ntiles][tile entries]  key integer << fb | flat position, bucket-sorted inside a tile
    int *toff;                      // [ntiles][nb + 1]  exclusive prefix of the tile's bucket counts
    int *cumul;                     // [nb][ntiles]      entries of bucket b in the tiles before t
    unsigned *where;                // [4 n_cap]         bucket << 14 | rank inside the (tile, bucket) run
    unsigned short *larr;           // [nb][maxe]        bucket-local vertex of arrival position i
    int *cursor;                    // [nb]              entries of bucket b
    int *wbase;                     // [nb]              first position of bucket b's window in `list` / `larr`
    int *big;                       // big[0] = buckets with more than maxe entries, big[1] = positions handed out in the overflow area
    int *biglist;                   // [BIGCAP]          those buckets (k_lat_bucket_big serves them: up to MAXE_BIG entries each)
    int *gcount;                    // [nb]              vertices of bucket b
    int4 *grec;                     // [nb][S]           (slot, start, length, first-seen flat position) of bucket-local vertex g
    int *gvix;                      // [nb][S]           vertex number of bucket-local vertex g
    int *vrec;                      // [h_cap]           b * S + g of vertex number h (the inverse of gvix)
    int4 *table;                    // [nb][S]           (key lo, key hi, vertex number, -) : the neighbour lookup
    unsigned *fbits;                // [W]               bit f = flat position f is the first-seen entry of its vertex
    int *wprefix;                   // [W]               set bits in front of word w inside its block of 2048 words
    int *bsum;                      // [W/2048]          set bits per block -> exclusive prefix
    int *ticket;                    // last-block election of k_lat_rank
    int nb, bb, S, sb, maxe;        // buckets = 1 << bb, slots per bucket = 1 << sb, entries per bucket
    int ntiles, tp;                 // tiles (capacity), points per tile
    int fb;                         // bits of a flat position (4 n_cap <= 1 << fb); a key integer must fit the other 64 - fb
    int use_big;                    // 0: k_lat_bucket_big is not launched - a bucket above maxe entries flags the level instead
    int want_off;                   // 0: lattice_offset is not needed (inference: the splat walks the vertex lists) - where / larr / cumul stay unwritten
};

__device__ __forceinline__ int part_bucket(uint64_t mixed, int bb) { return (int)(mixed >> (64 - bb)); }
__device__ __forceinline__ int part_slot(uint64_t mixed, int bb, int sb) { return (int)((mixed >> (64 - bb - sb)) & ((1u << sb) - 1u)); }

// ---- keys + barycentric weights of the partitioned build: k_point_keys with one extrema record per BLOCK of 256 points
// (sample, mins, maxs; sample -1: no points, -2: the block straddles a sample boundary)
__global__ void __launch_bounds__(TPB)
k_lat_keys(const float *__restrict__ pts, int64_t cstride, const int *__restrict__ n_dev, int n_cap, float scale32,
           float std32, float4 *__restrict__ bary, float4 *__restrict__ emg, int *__restrict__ part,
           const int *__restrict__ sid, int pps) {
    const int n = n_of(n_dev, n_cap);
    const int p = blockIdx.x * TPB + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int kmin[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
    int kmax[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    int blo = INT32_MAX, bhi = -1;
    if (p < n) {
        PointKeys pk;
        point_keys(pts[p], pts[cstride + p], pts[2 * cstride + p], scale32, std32, pk);
        bary[p] = make_float4(pk.bary[0], pk.bary[1], pk.bary[2], pk.bary[3]);
        emg[p] = make_float4(pk.emg[0], pk.emg[1], pk.emg[2], pk.emg[3]);
#pragma unroll
        for (int rem = 0; rem < 4; ++rem) {
            int k[4];
            entry_key(pk, rem, k);
#pragma unroll
            for (int c = 0; c < 4; ++c) { kmin[c] = min(kmin[c], k[c]); kmax[c] = max(kmax[c], k[c]); }
        }
        blo = bhi = sample_of(sid, pps, p);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        blo = min(blo, __shfl_xor(blo, o)); bhi = max(bhi, __shfl_xor(bhi, o));
#pragma unroll
        for (int c = 0; c < 4; ++c) { kmin[c] = min(kmin[c], __shfl_xor(kmin[c], o)); kmax[c] = max(kmax[c], __shfl_xor(kmax[c], o)); }
    }
    __shared__ int wrec[TPB / 64][10];
    if (lane == 0) {
        wrec[wv][0] = blo; wrec[wv][1] = bhi;
#pragma unroll
        for (int c = 0; c < 4; ++c) { wrec[wv][2 + c] = kmin[c]; wrec[wv][6 + c] = kmax[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        // word 0: the sample (or -1 / -2); words 1..4 mins, 5..8 maxs
        int lo = INT32_MAX, hi = -1;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) { lo = min(lo, wrec[w][0]); hi = max(hi, wrec[w][1]); }
        int v = 0;
        const int t = threadIdx.x;
        if (t == 0) v = hi < 0 ? -1 : (lo == hi ? lo : -2);
        else if (t <= 4) { v = INT32_MAX; for (int w = 0; w < TPB / 64; ++w) v = min(v, wrec[w][1 + t]); }
        else if (t <= 8) { v = INT32_MIN; for (int w = 0; w < TPB / 64; ++w) v = max(v, wrec[w][1 + t]); }
        part[12 * blockIdx.x + t] = v;
    }
}

// ---- per-sample key extrema from the per-block records of k_lat_keys: ONE workgroup, register folding + shuffle reductions +
// LDS atomics, plain stores of the result (no global atomics, nothing to initialise).  Blocks that straddle a sample boundary (at
// most one per boundary; record -2) are revisited point by point.
constexpr int MMT = 1024;
__global__ void __launch_bounds__(MMT)
k_lat_minmax(const int *__restrict__ part, const float *__restrict__ pts, int64_t cstride,
             const int *__restrict__ n_dev, int n_cap, float scale32, float std32, const int *__restrict__ sid, int pps,
             int nsamples, int *__restrict__ mm) {
    __shared__ int lmm[EFGH_LATTICE_MAX_SAMPLES * 8];
    __shared__ int strad[EFGH_LATTICE_MAX_SAMPLES];
    __shared__ int nstr;
    for (int i = threadIdx.x; i < nsamples * 8; i += MMT) lmm[i] = (i & 7) < 4 ? INT32_MAX : INT32_MIN;
    if (threadIdx.x == 0) nstr = 0;
    __syncthreads();
    const int n = n_of(n_dev, n_cap);
    const int nrec = (n + TPB - 1) / TPB;
    {
        // every thread folds a contiguous run of records (almost always one sample) in registers, four loads in flight; a change of
        // sample inside the run goes to the LDS table directly, the last one through a shuffle reduction per wave
        const int per = (nrec + MMT - 1) / MMT;
        int b = -1;
        int kmin[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
        int kmax[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
        for (int r0 = 0; r0 < per; r0 += 4) {
            int4 q0[4], q1[4], q2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = threadIdx.x * per + r0 + u;
                q0[u] = make_int4(-1, 0, 0, 0); q1[u] = q2[u] = q0[u];
                if (r0 + u < per && w < nrec) {
                    const int4 *q = reinterpret_cast<const int4 *>(part + 12 * w);
                    q0[u] = q[0]; q1[u] = q[1]; q2[u] = q[2];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (q0[u].x == -2) {
                    const int k = atomicAdd(&nstr, 1);
                    if (k < EFGH_LATTICE_MAX_SAMPLES) strad[k] = threadIdx.x * per + r0 + u;
                    continue;
                }
                if (q0[u].x < 0) continue;
                if (b >= 0 && q0[u].x != b) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { atomicMin(&lmm[8 * b + c], kmin[c]); atomicMax(&lmm[8 * b + 4 + c], kmax[c]); }
#pragma unroll
                    for (int c = 0; c < 4; ++c) { kmin[c] = INT32_MAX; kmax[c] = INT32_MIN; }
                }
                b = q0[u].x;
                kmin[0] = min(kmin[0], q0[u].y); kmin[1] = min(kmin[1], q0[u].z); kmin[2] = min(kmin[2], q0[u].w); kmin[3] = min(kmin[3], q1[u].x);
                kmax[0] = max(kmax[0], q1[u].y); kmax[1] = max(kmax[1], q1[u].z); kmax[2] = max(kmax[2], q1[u].w); kmax[3] = max(kmax[3], q2[u].x);
            }
        }
        wave_minmax_by_sample(b, kmin, kmax, lmm);
    }
    __syncthreads();
    const int ns = min(nstr, EFGH_LATTICE_MAX_SAMPLES);
    for (int k = threadIdx.x >> 6; k < ns * (TPB / 64); k += MMT / 64) {
        const int p = TPB * strad[k / (TPB / 64)] + 64 * (k % (TPB / 64)) + (threadIdx.x & 63);
        int b = -1;
        int kmin[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
        int kmax[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
        if (p < n) {
            PointKeys pk;
            point_keys(pts[p], pts[cstride + p], pts[2 * cstride + p], scale32, std32, pk);
#pragma unroll
            for (int rem = 0; rem < 4; ++rem) {
                int kk[4];
                entry_key(pk, rem, kk);
#pragma unroll
                for (int c = 0; c < 4; ++c) { kmin[c] = min(kmin[c], kk[c]); kmax[c] = max(kmax[c], kk[c]); }
            }
            b = sample_of(sid, pps, p);
        }
        wave_minmax_by_sample(b, kmin, kmax, lmm);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nsamples * 8; i += MMT) mm[i] = lmm[i];
}

// exclusive prefix over the NT threads of a block (NT = 256 or 512), one value each; *total = block sum
template <int NT>
__device__ __forceinline__ int block_exclusive_scan_n(int v, int *total) {
    __shared__ int wsum_n[NT / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum_n[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { if (i < w) base += wsum_n[i]; tot += wsum_n[i]; }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// ---- tile-local bucket sort of the entries.  A tile = STP * PPT points.
template <int PPT>
__global__ void __launch_bounds__(STP)
k_lat_scatter(const float *__restrict__ pts, int64_t cstride, const int *__restrict__ n_dev, int n_cap, float scale32,
              float std32, const int *__restrict__ mm, const int *__restrict__ sid, int pps, int nsamples, LatPart P,
              int *__restrict__ info) {
    __shared__ int cnt[NBMAX];
    const int n = n_of(n_dev, n_cap);
    const int p0 = blockIdx.x * (STP * PPT);
    if (p0 >= n) return;
    for (int b = threadIdx.x; b < P.nb; b += STP) cnt[b] = 0;
    __syncthreads();
    unsigned long long ki[PPT][4];
    unsigned br[PPT][4];                         // bucket << 14 | rank inside the (tile, bucket) run
    bool wide = false;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = p0 + j * STP + threadIdx.x;
        if (p < n) {
            PointKeys pk;
            point_keys(pts[p], pts[cstride + p], pts[2 * cstride + p], scale32, std32, pk);
            const int b = sample_of(sid, pps, p);
#pragma unroll
            for (int rem = 0; rem < 4; ++rem) {
                int k[4];
                entry_key(pk, rem, k);
                ki[j][rem] = (unsigned long long)(key2int(k, mm + 8 * b) * nsamples + b);
                wide |= (ki[j][rem] >> (64 - P.fb)) != 0ULL;
                const int bk = part_bucket(mix64(ki[j][rem]), P.bb);
                br[j][rem] = ((unsigned)bk << 14) | (unsigned)atomicAdd(&cnt[bk], 1);
            }
        }
    }
    if (wide) atomicOr(&info[EFGH_LATTICE_INFO_ERR], 4 | 8);    // key range x positions do not fit 64 bits: only the hash build serves the level (bit 3)
    __syncthreads();
    // exclusive prefix of the bucket counts: the tile's row of offsets, and the start of every run inside the tile's window
    {
        const int per = (P.nb + STP - 1) / STP;
        const int b0 = threadIdx.x * per;
        int mine = 0;
        for (int q = 0; q < per; ++q) if (b0 + q < P.nb) mine += cnt[b0 + q];
        int tot;
        int ex = block_exclusive_scan_n<STP>(mine, &tot);
        int *row = P.toff + (int64_t)blockIdx.x * (P.nb + 1);
        for (int q = 0; q < per; ++q) if (b0 + q < P.nb) {
            const int c = cnt[b0 + q];
            cnt[b0 + q] = ex;
            row[b0 + q] = ex;
            ex += c;
        }
        if (threadIdx.x == 0) row[P.nb] = tot;
    }
    __syncthreads();
    unsigned long long *win = P.ent + (int64_t)blockIdx.x * (4 * STP * PPT);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int p = p0 + j * STP + threadIdx.x;
        if (p < n) {
            uint4 w4;
            unsigned *wv = reinterpret_cast<unsigned *>(&w4);
#pragma unroll
            for (int rem = 0; rem < 4; ++rem) {
                const int bk = (int)(br[j][rem] >> 14);
                win[cnt[bk] + (int)(br[j][rem] & 0x3FFFu)] = (ki[j][rem] << P.fb) | (unsigned long long)(4u * (unsigned)p + rem);
                wv[rem] = br[j][rem];
            }
            if (P.want_off) reinterpret_cast<uint4 *>(P.where)[p] = w4;
        }
    }
}

// ---- one workgroup per bucket: gather its runs into LDS, group the entries by key there.  Nothing is carried in registers
// across the phases (the kernel is a chain of dependent LDS / memory round trips: what hides them is workgroups per CU).
template <int NT>
__device__ __forceinline__ int2 block_exclusive_scan2_n(int2 v, int2 *total) {
    __shared__ int2 wsum2_n[NT / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int2 x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y0 = __shfl_up(x.x, o), y1 = __shfl_up(x.y, o);
        if (lane >= o) { x.x += y0; x.y += y1; }
    }
    if (lane == 63) wsum2_n[w] = x;
    __syncthreads();
    int2 base = make_int2(0, 0), tot = make_int2(0, 0);
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) {
        if (i < w) { base.x += wsum2_n[i].x; base.y += wsum2_n[i].y; }
        tot.x += wsum2_n[i].x; tot.y += wsum2_n[i].y;
    }
    __syncthreads();
    *total = tot;
    return make_int2(base.x + x.x - v.x, base.y + x.y - v.y);
}

template <int MAXE, int BT, bool BIG>
__device__ __forceinline__ void
bucket_body(const LatPart &P, const int b, const int *__restrict__ n_dev, int n_cap, int *__restrict__ list, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];   // (64-bit LDS atomics on tkey: the dynamic area must not
                                                                              //  start at the odd end of the static one)
    const int S = P.S;
    unsigned long long *tkey = reinterpret_cast<unsigned long long *>(smem_);     // [S]
    unsigned long long *stage = tkey + S;                                         // [MAXE] entry words in arrival order; after the insert:
                                                                                  //        (slot << 16 | rank in slot) << 32 | flat position
    int *tcnt = reinterpret_cast<int *>(stage + MAXE);                            // [S]
    int *tstart = tcnt + S;                                                       // [S]
    int *tgid = tstart + S;                                                       // [S]
    int *lst = tgid + S;                                                          // [MAXE] flat positions, grouped by slot
    unsigned short *lslot = reinterpret_cast<unsigned short *>(lst + MAXE);       // [MAXE] slot of list position q
    // while the runs are gathered the list area holds the run table instead (NTMAX * 10 bytes <= MAXE * 6)
    int *runsrc = lst;                                                            // [NTMAX] source of tile t's run
    int *runpre = lst + NTMAX;                                                    // [NTMAX] arrival position of its head
    unsigned short *rid = reinterpret_cast<unsigned short *>(lst + 2 * NTMAX);    // [MAXE]  run of arrival position i
    static_assert(NTMAX * 8 + MAXE * 2 <= MAXE * 6, "run table must fit the list area");
    const int n = n_of(n_dev, n_cap);
    const int ntl = (n + P.tp - 1) / P.tp;                                        // tiles that hold points
    for (int s = threadIdx.x; s < S; s += BT) { tkey[s] = EMPTY; tcnt[s] = 0; }
    for (int i = threadIdx.x; i < MAXE; i += BT) rid[i] = 0;
    __syncthreads();
    int m, wbase;
    {
        const int te = 4 * P.tp;
        constexpr int PER = NTMAX / BT;
        int c[PER], mine = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int t = threadIdx.x * PER + q;
            c[q] = 0;
            if (t < ntl) {
                const int *row = P.toff + (int64_t)t * (P.nb + 1) + b;
                const int a0 = row[0];
                c[q] = row[1] - a0;
                runsrc[t] = t * te + a0;
            }
            mine += c[q];
        }
        int tot;
        int ex = block_exclusive_scan_n<BT>(mine, &tot);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int t = threadIdx.x * PER + q;
            if (t < ntl) {
                runpre[t] = ex;
                if (P.want_off) P.cumul[(int64_t)b * P.ntiles + t] = ex;
                if (c[q] > 0 && ex < MAXE) rid[ex] = (unsigned short)t;           // head of a non-empty run
            }
            ex += c[q];
        }
        __shared__ __attribute__((aligned(16))) int wb_s4[4];
        int &wb_s = wb_s4[0];
        if (tot > MAXE) {
            // more entries than this kernel's LDS holds (a lattice cell near the sensor of a real sweep collects thousands of
            // points): the bucket is handed to k_lat_bucket_big - or, beyond that kernel's capacity, flagged for the hash build
            if (threadIdx.x == 0) {
                P.cursor[b] = 0; P.gcount[b] = 0; P.wbase[b] = 0;
                bool listed = false;
                if (!BIG && P.use_big) {
                    const int k = atomicAdd(&P.big[0], 1);
                    if (k < BIGCAP) { P.biglist[k] = b; listed = true; }
                }
                if (!listed) atomicOr(&info[EFGH_LATTICE_INFO_ERR], 4);
            }
            return;
        }
        if (threadIdx.x == 0) {
            // normal buckets own window b of the list; the big ones are packed behind those windows
            wb_s = BIG ? P.nb * P.maxe + atomicAdd(&P.big[1], tot) : b * MAXE;
            P.cursor[b] = tot;
            P.wbase[b] = wb_s;
        }
        m = tot;
        __syncthreads();
        wbase = wb_s;
    }
    __syncthreads();
    // run of every arrival position: running maximum of the head marks (runs are laid out in tile order)
    {
        constexpr int PT = MAXE / BT;
        const int i0 = threadIdx.x * PT;
        int v[PT], mx = 0;
#pragma unroll
        for (int k = 0; k < PT; ++k) { mx = max(mx, (int)rid[i0 + k]); v[k] = mx; }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int x = mx;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x = max(x, y); }
        __shared__ int wmax[BT / 64];
        if (lane == 63) wmax[w] = x;
        __syncthreads();
        int before = __shfl_up(x, 1);
        if (lane == 0) before = 0;
#pragma unroll
        for (int q = 0; q < BT / 64; ++q) if (q < w) before = max(before, wmax[q]);
#pragma unroll
        for (int k = 0; k < PT; ++k) rid[i0 + k] = (unsigned short)max(v[k], before);
    }
    __syncthreads();
    for (int i0 = 0; i0 < m; i0 += 4 * BT) {
        unsigned long long wd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * BT + threadIdx.x;
            wd[u] = 0ULL;
            if (i < m) { const int t = rid[i]; wd[u] = P.ent[runsrc[t] + (i - runpre[t])]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * BT + threadIdx.x;
            if (i < m) stage[i] = wd[u];
        }
    }
    __syncthreads();
    constexpr int LONGL = 256, WGL = 1024;            // list lengths: counting sort up to LONGL, one wave's network up to WGL
    __shared__ int maxc_s;
    if (threadIdx.x == 0) maxc_s = 0;
    bool full = false;
    const unsigned long long fmask = (1ULL << P.fb) - 1ULL;
    for (int i = threadIdx.x; i < m; i += BT) {
        const unsigned long long wd = stage[i];
        const unsigned long long key = wd >> P.fb;
        int s = part_slot(mix64(key), P.bb, P.sb);
        int probe = 0;
        for (; probe < S; ++probe) {
            const unsigned long long prev = atomicCAS(&tkey[s], EMPTY, key);
            if (prev == EMPTY || prev == key) break;
            s = (s + 1) & (S - 1);
        }
        if (probe >= S) { full = true; s = 0; }
        const unsigned sr = ((unsigned)s << 16) | (unsigned)atomicAdd(&tcnt[s], 1);
        stage[i] = ((unsigned long long)sr << 32) | (wd & fmask);
    }
    if (full) atomicOr(&info[EFGH_LATTICE_INFO_ERR], 4);
    __syncthreads();
    // segments: exclusive scan of the slot counts (and of the occupied flags: the bucket-local vertex numbers)
    {
        const int per = (S + BT - 1) / BT;
        const int s0 = threadIdx.x * per;
        int2 mine = make_int2(0, 0);
        int cmax = 0;
        for (int q = 0; q < per; ++q) if (s0 + q < S) { const int c = tcnt[s0 + q]; mine.x += c; mine.y += c > 0; cmax = max(cmax, c); }
        if (cmax > LONGL) maxc_s = 1;                 // (some list of this bucket needs the sorting network)
        int2 tot;
        int2 ex = block_exclusive_scan2_n<BT>(mine, &tot);
        for (int q = 0; q < per; ++q) if (s0 + q < S) {
            const int c = tcnt[s0 + q];
            tstart[s0 + q] = ex.x; tgid[s0 + q] = ex.y;
            ex.x += c; ex.y += c > 0;
        }
        if (threadIdx.x == 0) P.gcount[b] = tot.y;
    }
    __syncthreads();
    unsigned short *larr = P.larr + wbase;
    for (int i = threadIdx.x; i < m; i += BT) {
        const unsigned long long v = stage[i];
        const unsigned sr = (unsigned)(v >> 32);
        const int s = (int)(sr >> 16);
        const int pos = tstart[s] + (int)(sr & 0xFFFFu);
        lst[pos] = (int)(unsigned)v;
        lslot[pos] = (unsigned short)s;
        if (P.want_off) larr[i] = (unsigned short)tgid[s];
    }
    __syncthreads();
    // Every vertex's list in ascending flat position (the splat's fixed summation order; its head is the vertex's first-seen entry).
    // Short lists: place of an entry = number of smaller positions in its list (positions are distinct) - neighbouring threads
    // hold neighbouring list positions, i.e. mostly the same vertex, so the count loop reads LDS as broadcasts; O(c^2), which is
    // what a real sweep punishes (cells near the sensor collect hundreds to thousands of points).  Lists longer than LONGL are
    // sorted in place first by a bitonic network whose compare-exchanges all point the same way (partner i ^ (k - 1) in the first
    // step of every merge), so a list of any length sorts as if padded with +inf: one wave per list up to WGL entries (a wave's LDS
    // operations execute in order: no barrier between the steps), the whole workgroup on the few longer ones.  Buckets without
    // a long list (all of them on the random-range bench scene) skip the block.
    if (maxc_s) {
        __shared__ int nlong_s, longs_s[64];              // the lists longer than WGL (at most MAXE / WGL <= 8 of them... 64 is generous)
        if (threadIdx.x == 0) nlong_s = 0;
        __syncthreads();
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int s = wv; s < S; s += BT / 64) {
            const int c = tcnt[s];
            if (c <= LONGL) continue;
            if (c > WGL) { if (lane == 0) { const int k = atomicAdd(&nlong_s, 1); if (k < 64) longs_s[k] = s; } continue; }
            int *a = lst + tstart[s];
            int p2 = 1;
            while (p2 < c) p2 <<= 1;
            for (int k = 2; k <= p2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int p = lane; p < (p2 >> 1); p += 64) {
                        const int lo = ((p / j) * (2 * j)) + (p % j);
                        const int hi = (j == (k >> 1)) ? (lo ^ (k - 1)) : (lo + j);
                        if (hi < c) { const int x = a[lo], y = a[hi]; if (x > y) { a[lo] = y; a[hi] = x; } }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
        }
        __syncthreads();
        const int nl = min(nlong_s, 64);
        for (int q = 0; q < nl; ++q) {
            const int s = longs_s[q], c = tcnt[s];
            int *a = lst + tstart[s];
            int p2 = 1;
            while (p2 < c) p2 <<= 1;
            for (int k = 2; k <= p2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int p = threadIdx.x; p < (p2 >> 1); p += BT) {
                        const int lo = ((p / j) * (2 * j)) + (p % j);
                        const int hi = (j == (k >> 1)) ? (lo ^ (k - 1)) : (lo + j);
                        if (hi < c) { const int x = a[lo], y = a[hi]; if (x > y) { a[lo] = y; a[hi] = x; } }
                    }
                    __syncthreads();
                }
        }
    }
    int *glist = list + wbase;
    for (int q = threadIdx.x; q < m; q += BT) {
        const int s = lslot[q];
        const int st = tstart[s], c = tcnt[s];
        const int me = lst[q];
        int r = q - st;                                   // (a sorted list: the place IS the rank)
        if (c <= LONGL) {
            r = 0;
            for (int k = 0; k < c; ++k) r += lst[st + k] < me ? 1 : 0;
        }
        glist[st + r] = me;
        if (r == 0) {
            atomicOr(&P.fbits[(unsigned)me >> 5], 1u << ((unsigned)me & 31u));
            P.grec[(int64_t)b * S + tgid[s]] = make_int4(s, st, c, me);
        }
    }
    int4 *tab = P.table + (int64_t)b * S;
    for (int s = threadIdx.x; s < S; s += BT) {
        const unsigned long long k = tkey[s];
        tab[s] = make_int4((int)(unsigned)k, (int)(unsigned)(k >> 32), -1, 0);
    }
}

template <int MAXE, int BT>
__global__ void __launch_bounds__(BT)
k_lat_bucket(LatPart P, const int *__restrict__ n_dev, int n_cap, int *__restrict__ list, int *__restrict__ info) {
    bucket_body<MAXE, BT, false>(P, blockIdx.x, n_dev, n_cap, list, info);
}

// the buckets k_lat_bucket left out (more than maxe entries): the same grouping with 152 KB of LDS and 1024 threads per bucket,
// windows handed out behind the regular ones.  Usually there are none: the launch then costs its latency only.
__global__ void __launch_bounds__(1024)
k_lat_bucket_big(LatPart P, const int *__restrict__ n_dev, int n_cap, int *__restrict__ list, int *__restrict__ info) {
    const int nbig = min(P.big[0], BIGCAP);
    for (int k = blockIdx.x; k < nbig; k += gridDim.x) {
        bucket_body<MAXE_BIG, 1024, true>(P, P.biglist[k], n_dev, n_cap, list, info);
        __syncthreads();
    }
}

// ---- first-seen numbering: vertex number = number of first-seen bits in front of the vertex's own.  Per 32-bit word its
// prefix inside a block of 2048 words (8 per thread: few blocks, so the last-block election costs few same-address atomics);
// the last block to finish turns the block sums into their exclusive prefix and writes H.
constexpr int RWT = 8;
__global__ void __launch_bounds__(TPB)
k_lat_rank(LatPart P, int W, int *__restrict__ info, int h_cap) {
    const int w0 = (blockIdx.x * TPB + threadIdx.x) * RWT;
    int c[RWT], mine = 0;
#pragma unroll
    for (int k = 0; k < RWT; ++k) { c[k] = w0 + k < W ? __popc(P.fbits[w0 + k]) : 0; mine += c[k]; }
    int tot;
    int ex = block_exclusive_scan(mine, &tot);
#pragma unroll
    for (int k = 0; k < RWT; ++k) { if (w0 + k < W) P.wprefix[w0 + k] = ex; ex += c[k]; }
    __shared__ int last_s;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&P.bsum[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last_s = atomicAdd(P.ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    const int nb = gridDim.x;
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int s = 0; s < nb; s += TPB) {
        const int i = s + threadIdx.x;
        const int v = i < nb ? __hip_atomic_load(&P.bsum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int t2;
        const int e2 = block_exclusive_scan(v, &t2);
        const int carry = carry_s;
        if (i < nb) P.bsum[i] = carry + e2;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + t2;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        info[EFGH_LATTICE_INFO_H] = carry_s;
        if (carry_s > h_cap) atomicOr(&info[EFGH_LATTICE_INFO_ERR], 1);
    }
}

// ---- number the vertices: one thread per (bucket, bucket-local vertex) slot of the grid nb x S.  Only the number leaves this
// kernel (to the bucket's tables and, as the inverse vrec[number] = (bucket, vertex), to k_lat_nbr, which emits the vertex
// records in number order with coalesced stores).
__global__ void __launch_bounds__(TPB)
k_lat_number(LatPart P, int h_cap) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    const int b = (int)(i >> P.sb), g = (int)(i & (P.S - 1));
    if (b >= P.nb || g >= P.gcount[b]) return;
    const int4 rec = P.grec[i];
    const unsigned hf = (unsigned)rec.w;
    const unsigned w = hf >> 5;
    const int idx = P.bsum[w >> 11] + P.wprefix[w] + __popc(P.fbits[w] & ((1u << (hf & 31u)) - 1u));
    P.gvix[i] = idx;
    if (idx >= h_cap) return;
    P.table[(int64_t)b * P.S + rec.x].z = idx;
    P.vrec[idx] = (int)i;
}

struct LatGeom {                    // what the point -> key recipe needs
    const float *pts; int64_t cstride; float scale32, std32, div32; const int *sid; int pps;
};

// key, sample of vertex h from its first-seen entry
__device__ __forceinline__ void vertex_key(const LatPart &P, const LatGeom &G, int h, int k[4], int &smp, int4 &rec, int &p, int &rem) {
    rec = P.grec[P.vrec[h]];
    p = (int)((unsigned)rec.w >> 2); rem = rec.w & 3;
    PointKeys pk;
    point_keys(G.pts[p], G.pts[G.cstride + p], G.pts[2 * G.cstride + p], G.scale32, G.std32, pk);
    entry_key(pk, rem, k);
    smp = sample_of(G.sid, G.pps, p);
}

// ---- one launch, three independent jobs selected by the block index:
//   blocks [0, nbr_blocks)      15 blur neighbours per vertex through the buckets' table images (see k_neighbors for the aliasing
//                               semantics); the vertex's key is recomputed from its first-seen point by lane t = 0 of its 16 lanes
//   the next off_blocks         lattice_offset per point, gathered through where[] -> arrival position -> bucket-local vertex ->
//                               vertex number (three loads, two of them into L2-resident tables) to a coalesced 16-byte store
//   the rest                    vertex records in number order: list segment, sample, the next level's point
__global__ void __launch_bounds__(TPB)
k_lat_nbr(LatPart P, LatGeom G, const int *__restrict__ mm, int *__restrict__ info, int h_cap, int *__restrict__ nbr,
          int nsamples, int2 *__restrict__ alist, int alias_cap, int nbr_blocks, int off_blocks, const int *__restrict__ n_dev,
          int n_cap, int4 *__restrict__ off, int2 *__restrict__ vseg, float *__restrict__ pts_next, int h_cap_build,
          int *__restrict__ vsid) {
    int H = info[EFGH_LATTICE_INFO_H];
    if (H > h_cap) H = h_cap;
    if ((int)blockIdx.x >= nbr_blocks + off_blocks) {
        for (int h = (blockIdx.x - nbr_blocks - off_blocks) * TPB + threadIdx.x; h < H; h += (gridDim.x - nbr_blocks - off_blocks) * TPB) {
            int k[4], smp, p, rem;
            int4 rec;
            vertex_key(P, G, h, k, smp, rec, p, rem);
            vseg[h] = make_int2(P.wbase[P.vrec[h] >> P.sb] + rec.y, rec.z);
            vsid[h] = smp;
            // vertices are numbered sample-major, and the very first key of a sample is always new: its number is the sample's first
            if (rem == 0 && (p == 0 || sample_of(G.sid, G.pps, p - 1) != smp)) info[EFGH_LATTICE_INFO_SEG + smp] = h;
            // generate_data.py:176-178: key (as fp32) / float32(std*scale), then E^T . (4-term fma chain)
            float kf[4] = {__fdiv_rn((float)k[0], G.div32), __fdiv_rn((float)k[1], G.div32),
                           __fdiv_rn((float)k[2], G.div32), __fdiv_rn((float)k[3], G.div32)};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float acc = __fmul_rn(elev(0, q), kf[0]);
                acc = __fmaf_rn(elev(1, q), kf[1], acc);
                acc = __fmaf_rn(elev(2, q), kf[2], acc);
                acc = __fmaf_rn(elev(3, q), kf[3], acc);
                pts_next[(int64_t)q * h_cap_build + h] = acc;
            }
        }
        return;
    }
    if ((int)blockIdx.x >= nbr_blocks) {
        const int n = n_of(n_dev, n_cap);
        const int p = (blockIdx.x - nbr_blocks) * TPB + threadIdx.x;
        if (p >= n) return;
        const uint4 w4 = reinterpret_cast<const uint4 *>(P.where)[p];
        const unsigned wv[4] = {w4.x, w4.y, w4.z, w4.w};
        const int t = p / P.tp;
        int o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bk = (int)(wv[r] >> 14);
            const int pos = P.cumul[(int64_t)bk * P.ntiles + t] + (int)(wv[r] & 0x3FFFu);
            // (a flagged bucket has window 0 and whatever lies there: the mask keeps the lookup inside the bucket's own vertices)
            o[r] = P.gvix[(int64_t)bk * P.S + (P.larr[P.wbase[bk] + pos] & (P.S - 1))];
        }
        off[p] = make_int4(o[0], o[1], o[2], o[3]);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int S = P.S;
    for (int64_t g0 = (int64_t)blockIdx.x * TPB; g0 < (int64_t)H * 16; g0 += (int64_t)nbr_blocks * TPB) {
        const int64_t g = g0 + threadIdx.x;
        const int h = (int)(g >> 4), t = (int)(g & 15);
        int k0[4] = {0, 0, 0, 0}, b = 0;
        if (h < H && t == 0) {
            int p_, rem_;
            int4 rec_;
            vertex_key(P, G, h, k0, b, rec_, p_, rem_);
        }
        const int src = lane & 48;               // lane t = 0 of this vertex's 16 lanes
        b = __shfl(b, src);
#pragma unroll
        for (int c = 0; c < 4; ++c) k0[c] = __shfl(k0[c], src);
        int res = -1;
        bool aliased = false;
        if (h < H && t == 0) res = h;              // offset 0: the vertex itself
        if (h < H && t > 0 && t < 15) {
            int k[4] = {k0[0] + c_nbr[t][0], k0[1] + c_nbr[t][1], k0[2] + c_nbr[t][2], k0[3] + c_nbr[t][3]};
            const int *m8 = mm + 8 * b;
            int64_t ki = key2int(k, m8);
            if (ki >= 0) {   // every inserted key integer is >= 0
                ki = ki * nsamples + b;
                const uint64_t mixed = mix64((uint64_t)ki);
                const int4 *tab = P.table + (int64_t)part_bucket(mixed, P.bb) * S;
                int s = part_slot(mixed, P.bb, P.sb);
                const int klo = (int)(unsigned)(uint64_t)ki, khi = (int)(unsigned)((uint64_t)ki >> 32);
                for (int probe = 0; probe < S; ++probe) {
                    const int4 cur = tab[s];
                    if (cur.x == -1 && cur.y == -1) break;           // EMPTY
                    if (cur.x == klo && cur.y == khi) { res = cur.z; break; }
                    s = (s + 1) & (S - 1);
                }
            }
            aliased = res >= 0 && (k[1] < m8[1] || k[1] > m8[5] || k[2] < m8[2] || k[2] > m8[6] || k[3] < m8[3] || k[3] > m8[7]);
        }
        const unsigned long long am = __ballot(aliased);
        if (h < H) {
            if (t < 15) nbr[g] = res;
            else nbr[g] = (int)((am >> (lane & 48)) & 0x7FFFu);
        }
        if (am) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&info[EFGH_LATTICE_INFO_ALIAS], __popcll(am));
            base = __shfl(base, 0);
            if (aliased) {
                const int k_ = base + __popcll(am & ((1ULL << lane) - 1ULL));
                if (k_ < alias_cap) alist[k_] = make_int2((int)g, res);
                else atomicOr(&info[EFGH_LATTICE_INFO_ERR], 2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The TAIL of the pyramid as ONE launch (round 5).  The lattices of different samples never interact - own key extrema, own
// keys, own first-seen numbering - so once a sample's level fits one workgroup's LDS (<= 8 192 points: levels 3 and 4 of the
// bench scene, every level of a small cloud) the whole level is built by ONE workgroup with __syncthreads() only: keys +
// extrema, an open-addressing table int64 key -> (first-seen position, count) in LDS, first-seen numbering as a prefix
// popcount over one bit per flat position, vertex records, lattice_offset, the ascending entry lists and the 15 neighbour probes
// against the same LDS table.  The level's vertices ARE the sample's points of the next level, so the workgroup walks down the
// remaining levels itself.  The only thing that crosses samples is the sample-major vertex base (an exclusive prefix of B
// counts): the B co-resident workgroups meet once per level on a ticket in the level's zeroed info block (B <= 64, one
// workgroup per CU: co-residency is guaranteed on 256 CUs whatever else runs).  Replaces 7 launches per level, each at its
// 5-12 us floor (profiles/r04_bcl_timeline.txt): 21 launches -> 1 for levels 2-4 worth of floors.
// Anything it cannot hold (a sample with more points, more vertices than 0.8 S slots, a list longer than TAIL_MAX_LIST) sets bit 2
// of the level's ERR word: efgh_amd/lattice.py then rebuilds the pyramid with the per-level kernels.
constexpr int TAIL_THREADS = 1024;
constexpr int TAIL_NMAX = 8192;           // points of one sample and level (32 768 flat positions = 1 024 bitmap words)
constexpr int TAIL_MAX_LIST = 2048;       // entries of one vertex (the rank sort below is quadratic in it)
constexpr int TAIL_MAX_LEVELS = 5;

struct TailLevel {
    float scale32, div32; int h_cap, alias_cap;
    float4 *bary, *emg; int4 *off; int *list; int2 *vseg; int *nbr; float *pts_next; int *vsid; int *info; int2 *alist;
};
struct TailArgs {
    TailLevel lv[TAIL_MAX_LEVELS];
    int nlevels, nsamples, S, pps;        // pps: points per sample when the first tail level is level 0 (info_prev == NULL)
    const float *pts; int64_t cstride; const int *info_prev; int prev_h_cap; float std32;
};

// exclusive prefix of one value per thread over the workgroup (1 024 threads); every thread gets the total
__device__ __forceinline__ int tail_scan(int v, int *wsum, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    __syncthreads();                       // (wsum may still be read from a previous call)
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TAIL_THREADS / 64; ++w) { const int x = wsum[w]; tot += x; if (w < wave) base += x; }
    total = tot;
    return base + inc - v;
}

__device__ __forceinline__ int tail_probe(const unsigned long long *tkey, int S, long long ki) {
    int s = (int)(mix64((uint64_t)ki) >> 20) & (S - 1);
    for (int probe = 0; probe < S; ++probe) {
        const unsigned long long cur = tkey[s];
        if (cur == (unsigned long long)ki) return s;
        if (cur == EMPTY) return -1;
        s = (s + 1) & (S - 1);
    }
    return -1;
}

__global__ void __launch_bounds__(TAIL_THREADS, 1) k_lat_tail(const TailArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    const int S = A.S, tid = threadIdx.x, lane = tid & 63, b = blockIdx.x, B = A.nsamples;
    unsigned long long *tkey = reinterpret_cast<unsigned long long *>(tail_smem);      // [S] key integer of the slot
    int *tmin = reinterpret_cast<int *>(tkey + S);       // [S] smallest local flat position of the key (its first-seen entry)
    int *tcnt = tmin + S;                                // [S] entries of the key; later the placement cursor
    int *tnum = tcnt + S;                                // [S] local vertex number of the slot
    int *vfirst = tnum + S;                              // [S] by local vertex number: first-seen local flat position
    int *vcnt = vfirst + S;                              // [S] by local vertex number: entries
    int *vstart = vcnt + S;                              // [S + 4] by local vertex number: list start (exclusive scan); first the word prefix
    unsigned *bits = reinterpret_cast<unsigned *>(vstart + S + 4);      // [1024] one bit per local flat position: first-seen entries
    int *mm = reinterpret_cast<int *>(bits + 1024);      // [8] key extrema of the sample
    int *wsum = mm + 8;                                  // [16]
    int *misc = wsum + 16;                               // [0] overflow, [1] base, [2] total, [3] longest list
    unsigned short *stage = reinterpret_cast<unsigned short *>(misc + 4);      // [4 * TAIL_NMAX] the lists in arrival order (local flat positions)

    const float *pts = A.pts;
    int64_t cs = A.cstride;
    int p0, n;
    if (A.info_prev) {
        const int hp = min(A.info_prev[EFGH_LATTICE_INFO_H], A.prev_h_cap);
        p0 = A.info_prev[EFGH_LATTICE_INFO_SEG + b];
        const int e = b + 1 < B ? A.info_prev[EFGH_LATTICE_INFO_SEG + b + 1] : hp;
        n = e - p0;
        if ((A.info_prev[EFGH_LATTICE_INFO_ERR] & 5) || n < 0) n = 0;       // (the level above is garbage: flagged there)
    } else { p0 = b * A.pps; n = A.pps; }

    for (int l = 0; l < A.nlevels; ++l) {
        const TailLevel &L = A.lv[l];
        const bool fit_in = n <= TAIL_NMAX;
        const int nn = fit_in ? n : 0;                   // points this workgroup really processes
        // ---- 0: clear
        for (int i = tid; i < S; i += TAIL_THREADS) { tkey[i] = EMPTY; tmin[i] = INT32_MAX; tcnt[i] = 0; }
        bits[tid] = 0u;
        if (tid < 8) mm[tid] = tid < 4 ? INT32_MAX : INT32_MIN;
        if (tid < 4) misc[tid] = 0;
        __syncthreads();
        if (!fit_in && tid == 0) misc[0] = 1;
        // ---- 1: barycentric weights, el_minus_gr, key extrema
        {
            int kmin[4] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX}, kmax[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
            for (int i = tid; i < nn; i += TAIL_THREADS) {
                const int p = p0 + i;
                PointKeys pk;
                point_keys(pts[p], pts[cs + p], pts[2 * cs + p], L.scale32, A.std32, pk);
                L.bary[p] = make_float4(pk.bary[0], pk.bary[1], pk.bary[2], pk.bary[3]);
                L.emg[p] = make_float4(pk.emg[0], pk.emg[1], pk.emg[2], pk.emg[3]);
#pragma unroll
                for (int rem = 0; rem < 4; ++rem) {
                    int k[4];
                    entry_key(pk, rem, k);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { kmin[c] = min(kmin[c], k[c]); kmax[c] = max(kmax[c], k[c]); }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { kmin[c] = min(kmin[c], __shfl_xor(kmin[c], o)); kmax[c] = max(kmax[c], __shfl_xor(kmax[c], o)); }
            }
            if (lane == 0 && nn > 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { atomicMin(&mm[c], kmin[c]); atomicMax(&mm[4 + c], kmax[c]); }
            }
        }
        __syncthreads();
        // ---- 2: insert (key integer -> slot; smallest flat position and count per slot)
        for (int i = tid; i < nn; i += TAIL_THREADS) {
            const int p = p0 + i;
            PointKeys pk;
            point_keys(pts[p], pts[cs + p], pts[2 * cs + p], L.scale32, A.std32, pk);
#pragma unroll
            for (int rem = 0; rem < 4; ++rem) {
                int k[4];
                entry_key(pk, rem, k);
                const unsigned long long ki = (unsigned long long)key2int(k, mm);
                int s = (int)(mix64(ki) >> 20) & (S - 1), probe = 0;
                for (; probe < S; ++probe) {
                    unsigned long long cur = tkey[s];
                    if (cur == EMPTY) cur = atomicCAS(&tkey[s], EMPTY, ki);
                    if (cur == EMPTY || cur == ki) break;
                    s = (s + 1) & (S - 1);
                }
                if (probe >= S) { misc[0] = 1; continue; }
                atomicMin(&tmin[s], 4 * i + rem);
                atomicAdd(&tcnt[s], 1);
            }
        }
        __syncthreads();
        // ---- 3: first-seen numbering = number of first-seen bits in front of the vertex's own
        for (int s = tid; s < S; s += TAIL_THREADS)
            if (tkey[s] != EMPTY) { const int f = tmin[s]; atomicOr(&bits[f >> 5], 1u << (f & 31)); atomicMax(&misc[3], tcnt[s]); }
        __syncthreads();
        int Hb;
        {
            const int pre = tail_scan(__popc(bits[tid]), wsum, Hb);
            vstart[tid] = pre;                           // (word prefix, S >= 1024)
        }
        __syncthreads();
        if (tid == 0 && (Hb * 10 > S * 8 || misc[3] > TAIL_MAX_LIST)) misc[0] = 1;
        for (int s = tid; s < S; s += TAIL_THREADS)
            if (tkey[s] != EMPTY) {
                const int f = tmin[s];
                const int num = vstart[f >> 5] + __popc(bits[f >> 5] & ((1u << (f & 31)) - 1u));
                tnum[s] = num; vfirst[num] = f; vcnt[num] = tcnt[s];
            }
        __syncthreads();
        const bool over = misc[0] != 0;
        if (over) Hb = 0;
        // ---- 4: the sample-major vertex base: the B workgroups of the launch meet on the level's ticket
        if (tid == 0) {
            int *slot = L.info + EFGH_LATTICE_INFO_SEG + B + 1;
            __hip_atomic_store(&slot[b], Hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            __hip_atomic_fetch_add(&L.info[EFGH_LATTICE_INFO_SEG + B], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&L.info[EFGH_LATTICE_INFO_SEG + B], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < B) __builtin_amdgcn_s_sleep(2);
            __threadfence();
            int base = 0, tot = 0;
            for (int q = 0; q < B; ++q) {
                const int h = __hip_atomic_load(&slot[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q < b) base += h;
                tot += h;
            }
            misc[1] = base; misc[2] = tot;
            L.info[EFGH_LATTICE_INFO_SEG + b] = base;
            if (over) atomicOr(&L.info[EFGH_LATTICE_INFO_ERR], 4);
            if (b == 0) {
                L.info[EFGH_LATTICE_INFO_H] = tot;
                if (tot > L.h_cap) atomicOr(&L.info[EFGH_LATTICE_INFO_ERR], 1);
            }
        }
        __syncthreads();
        const int base = misc[1], Htot = misc[2];
        const bool ok = !over && Htot <= L.h_cap;        // (else nothing vertex-indexed is written; the level is flagged)
        // ---- 5: list starts (exclusive scan of the counts in vertex order), vertex records
        {
            constexpr int IT = 4;                        // S <= 4096: four consecutive vertices per thread
            int c[IT], sum = 0;
#pragma unroll
            for (int q = 0; q < IT; ++q) { const int v = tid * IT + q; c[q] = (v < Hb && v < S) ? vcnt[v] : 0; sum += c[q]; }
            int tot;
            int pre = tail_scan(sum, wsum, tot);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < IT; ++q) { const int v = tid * IT + q; if (v <= Hb && v < S + 1) vstart[v] = pre; pre += c[q]; }
        }
        __syncthreads();
        if (ok) {
            for (int v = tid; v < Hb; v += TAIL_THREADS) {
                const int f = vfirst[v], p = p0 + (f >> 2), rem = f & 3;
                PointKeys pk;
                point_keys(pts[p], pts[cs + p], pts[2 * cs + p], L.scale32, A.std32, pk);
                int k[4];
                entry_key(pk, rem, k);
                const int h = base + v;
                L.vseg[h] = make_int2(4 * p0 + vstart[v], vcnt[v]);
                L.vsid[h] = b;
                const float kf[4] = {__fdiv_rn((float)k[0], L.div32), __fdiv_rn((float)k[1], L.div32),
                                     __fdiv_rn((float)k[2], L.div32), __fdiv_rn((float)k[3], L.div32)};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    float acc = __fmul_rn(elev(0, q), kf[0]);
                    acc = __fmaf_rn(elev(1, q), kf[1], acc);
                    acc = __fmaf_rn(elev(2, q), kf[2], acc);
                    acc = __fmaf_rn(elev(3, q), kf[3], acc);
                    L.pts_next[(int64_t)q * L.h_cap + h] = acc;
                }
            }
        }
        // ---- 6: lattice_offset per point; every entry into its vertex's list (arrival order), then a rank sort per entry
        for (int s = tid; s < S; s += TAIL_THREADS) tcnt[s] = 0;          // (placement cursors)
        __syncthreads();
        int ef[TAIL_NMAX / TAIL_THREADS][4], ev[TAIL_NMAX / TAIL_THREADS][4];
#pragma unroll
        for (int it = 0; it < TAIL_NMAX / TAIL_THREADS; ++it) {
            const int i = it * TAIL_THREADS + tid;
#pragma unroll
            for (int rem = 0; rem < 4; ++rem) { ef[it][rem] = -1; ev[it][rem] = 0; }
            if (i < nn && ok) {
                const int p = p0 + i;
                PointKeys pk;
                point_keys(pts[p], pts[cs + p], pts[2 * cs + p], L.scale32, A.std32, pk);
                int o[4];
#pragma unroll
                for (int rem = 0; rem < 4; ++rem) {
                    int k[4];
                    entry_key(pk, rem, k);
                    const int s = tail_probe(tkey, S, key2int(k, mm));
                    const int v = s >= 0 ? tnum[s] : 0;
                    o[rem] = base + v;
                    if (s >= 0) {
                        const int pos = vstart[v] + atomicAdd(&tcnt[s], 1);
                        stage[pos] = (unsigned short)(4 * i + rem);
                        ef[it][rem] = 4 * i + rem; ev[it][rem] = v;
                    }
                }
                if (L.off) L.off[p] = make_int4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
        // rank of every entry inside its vertex's list (lists are short: counting in LDS), then ONE store to its sorted place
#pragma unroll
        for (int it = 0; it < TAIL_NMAX / TAIL_THREADS; ++it)
#pragma unroll
            for (int rem = 0; rem < 4; ++rem) {
                const int f = ef[it][rem];
                if (f < 0) continue;
                const int v = ev[it][rem], j0 = vstart[v], j1 = vstart[v + 1];
                int r = 0;
                for (int j = j0; j < j1; ++j) r += (int)stage[j] < f ? 1 : 0;
                L.list[4 * p0 + j0 + r] = 4 * p0 + f;
            }
        // ---- 7: the 15 blur neighbours per vertex (16 lanes per vertex; key2int without a range check, aliased hits recorded)
        if (ok) {
            for (int g0 = 0; g0 < Hb * 16; g0 += TAIL_THREADS) {
                const int g = g0 + tid, v = g >> 4, t = g & 15;
                int k0[4] = {0, 0, 0, 0};
                if (v < Hb && t == 0) {
                    const int f = vfirst[v], p = p0 + (f >> 2);
                    PointKeys pk;
                    point_keys(pts[p], pts[cs + p], pts[2 * cs + p], L.scale32, A.std32, pk);
                    entry_key(pk, f & 3, k0);
                }
                const int src = lane & 48;
#pragma unroll
                for (int c = 0; c < 4; ++c) k0[c] = __shfl(k0[c], src);
                int res = -1;
                bool aliased = false;
                if (v < Hb && t == 0) res = base + v;
                if (v < Hb && t > 0 && t < 15) {
                    const int k[4] = {k0[0] + c_nbr[t][0], k0[1] + c_nbr[t][1], k0[2] + c_nbr[t][2], k0[3] + c_nbr[t][3]};
                    const int64_t ki = key2int(k, mm);
                    if (ki >= 0) { const int s = tail_probe(tkey, S, ki); if (s >= 0) res = base + tnum[s]; }
                    aliased = res >= 0 && (k[1] < mm[1] || k[1] > mm[5] || k[2] < mm[2] || k[2] > mm[6] || k[3] < mm[3] || k[3] > mm[7]);
                }
                const unsigned long long am = __ballot(aliased);
                const int64_t gg = (int64_t)(base + v) * 16 + t;
                if (v < Hb) L.nbr[gg] = t < 15 ? res : (int)((am >> (lane & 48)) & 0x7FFFu);
                if (am) {
                    int ab = 0;
                    if (lane == 0) ab = atomicAdd(&L.info[EFGH_LATTICE_INFO_ALIAS], __popcll(am));
                    ab = __shfl(ab, 0);
                    if (aliased) {
                        const int k_ = ab + __popcll(am & ((1ULL << lane) - 1ULL));
                        if (k_ < L.alias_cap) L.alist[k_] = make_int2((int)gg, res);
                        else atomicOr(&L.info[EFGH_LATTICE_INFO_ERR], 2);
                    }
                }
            }
        }
        // ---- the level's vertices are this sample's points of the next level
        __threadfence();
        __syncthreads();
        __threadfence();
        pts = L.pts_next; cs = L.h_cap; p0 = base; n = ok ? Hb : 0;
    }
}

int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

struct WsLayout {
    int64_t slot, rnk, list0, flags, bsum, hkeys, cnt, sstart, hvals, mm, vkeys, part, bsum2, occ, total;
};

WsLayout ws_layout(int32_t n_cap, int32_t h_cap, int32_t nsamples, int64_t hcap) {
    WsLayout w;
    int64_t o = 0;
    w.slot = o;   o += align256((int64_t)n_cap * 16);
    w.rnk = o;    o += align256((int64_t)n_cap * 16);
    w.list0 = o;  o += align256((int64_t)n_cap * 16);
    w.flags = o;  o += align256((int64_t)n_cap * 4);
    w.bsum = o;   o += align256(((int64_t)cdiv(n_cap, TPB) + 64) * 4);
    w.hkeys = o;  o += align256(hcap * 8);
    w.cnt = o;    o += align256(hcap * 4);
    w.sstart = o; o += align256(hcap * 4);
    w.hvals = o;  o += align256(hcap * 4);
    w.mm = o;     o += align256((int64_t)nsamples * 32);
    w.vkeys = o;  o += align256((int64_t)h_cap * 16);
    w.part = o;   o += align256(((int64_t)cdiv(n_cap, 64) + 4) * 48);
    w.bsum2 = o;  o += align256((hcap / 1024 + 1) * 8);
    w.occ = o;    o += align256((int64_t)n_cap * 64);
    w.total = o;
    return w;
}

}  // namespace

extern "C" int64_t efgh_lattice_hash_capacity(int32_t n_in) {
    int64_t c = 4096;
    while (c < (int64_t)n_in * 8) c <<= 1;
    return c;
}

extern "C" int64_t efgh_lattice_workspace_bytes(int32_t n_cap, int32_t h_cap, int32_t nsamples) {
    return ws_layout(n_cap, h_cap, nsamples, efgh_lattice_hash_capacity(n_cap)).total;
}

// hash_slots = 0: the default table (no overflow possible); else a power of two in [4096, default]
static int64_t pick_hash_slots(int32_t n_cap, int64_t hash_slots) {
    const int64_t dflt = efgh_lattice_hash_capacity(n_cap);
    if (hash_slots <= 0 || hash_slots >= dflt) return dflt;
    if (hash_slots < 4096 || (hash_slots & (hash_slots - 1))) return -1;
    return hash_slots;
}

extern "C" int efgh_lattice_level_build(const float *pts, int64_t pts_cstride, const int32_t *n_dev, int32_t n_cap,
                                        const int32_t *sid, int32_t pts_per_sample, int32_t nsamples, float scale32,
                                        float div32, float *bary, float *emg, int32_t *off, int32_t *list, int32_t h_cap,
                                        int32_t *vseg, float *pts_next, int32_t *vsid, int32_t *info, void *workspace,
                                        int64_t hash_slots, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(n_cap > 0 && n_cap < (1 << 27) && h_cap > 0 && nsamples >= 1 && nsamples <= EFGH_LATTICE_MAX_SAMPLES);      // (LDS [nsamples][8] in k_minmax_finalize)
    EFGH_CHECK_ARG(sid || pts_per_sample > 0);
    EFGH_CHECK_ARG(pts && bary && emg && off && list && vseg && pts_next && vsid && info && workspace);
    const int64_t hcap = pick_hash_slots(n_cap, hash_slots);
    EFGH_CHECK_ARG(hcap > 0);
    const WsLayout w = ws_layout(n_cap, h_cap, nsamples, hcap);
    char *ws = (char *)workspace;
    int4 *slot = (int4 *)(ws + w.slot), *rnk = (int4 *)(ws + w.rnk);
    int *list0 = (int *)(ws + w.list0), *flags = (int *)(ws + w.flags), *bsum = (int *)(ws + w.bsum);
    unsigned long long *hkeys = (unsigned long long *)(ws + w.hkeys);
    int *cnt = (int *)(ws + w.cnt), *sstart = (int *)(ws + w.sstart), *hvals = (int *)(ws + w.hvals), *mm = (int *)(ws + w.mm);
    int4 *vkeys = (int4 *)(ws + w.vkeys);
    int *part = (int *)(ws + w.part);
    int4 *occ = (int4 *)(ws + w.occ);
    int2 *bsum2 = (int2 *)(ws + w.bsum2);
    const uint32_t std_bits = 0x405105ECu;         // float32(4*sqrt(2/3)), generate_data.py:19
    float std32;
    memcpy(&std32, &std_bits, 4);
    const int nbp = cdiv(n_cap, TPB);
    const int pps = sid ? 1 : pts_per_sample;
    int ginit = cdiv(hcap, TPB * 4);
    if (ginit > 8192) ginit = 8192;
    k_level_init<<<ginit, TPB, 0, st>>>(hkeys, cnt, hcap, flags, n_cap, mm, nsamples, info, EFGH_LATTICE_INFO_SEG + nsamples);
    k_point_keys<<<nbp, TPB, 0, st>>>(pts, pts_cstride, n_dev, n_cap, scale32, std32, (float4 *)bary, (float4 *)emg, mm, part, sid,
                                      pps);
    k_minmax_finalize<<<cdiv(nbp * (TPB / 64), TPB), TPB, 0, st>>>(part, nbp * (TPB / 64), mm);
    k_insert<<<nbp, TPB, 0, st>>>(pts, pts_cstride, n_dev, n_cap, scale32, std32, mm, hkeys, cnt, hcap - 1, slot, rnk, sid, pps,
                                  nsamples, info);
    const int nb2 = (int)(hcap / (1024 * SEG_Q)) > 0 ? (int)(hcap / (1024 * SEG_Q)) : 1;      // hcap is a power of two >= 4096
    k_seg_count<<<nb2, TPB, 0, st>>>((const int4 *)cnt, bsum2);
    k_seg_scan<<<1, TPB, 0, st>>>(bsum2, nb2, info + EFGH_LATTICE_INFO_CURSOR);
    k_seg_assign<<<nb2, TPB, 0, st>>>((const int4 *)cnt, bsum2, (int4 *)sstart, occ);
    k_place<<<nbp, TPB, 0, st>>>(slot, rnk, n_dev, n_cap, sstart, list0);
    {
        int g = cdiv((int64_t)n_cap * 4 * 32, TPB);           // at most 4*n_cap occupied slots, a half-wave each
        if (g > 16384) g = 16384;
        k_sortmin<<<g, TPB, 0, st>>>(occ, info + EFGH_LATTICE_INFO_CURSOR, list0, list, (unsigned char *)flags);
    }
    k_flag_count<<<nbp, TPB, 0, st>>>((const unsigned *)flags, n_dev, n_cap, bsum);
    k_scan_sums<<<1, TPB, 0, st>>>(bsum, nbp, info, h_cap);
    k_assign<<<nbp, TPB, 0, st>>>(pts, pts_cstride, n_dev, n_cap, scale32, std32, (const unsigned *)flags, slot, bsum, cnt, sstart,
                                  hvals, vkeys, (int2 *)vseg, pts_next, h_cap, div32, sid, pps, vsid, info);
    k_offsets<<<nbp, TPB, 0, st>>>(slot, hvals, n_dev, n_cap, (int4 *)off);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_lattice_level_neighbors(const void *workspace, int32_t n_cap, int32_t h_cap_build, int32_t nsamples,
                                            int32_t *info, const int32_t *vsid, int32_t h_cap, int32_t *nbr,
                                            int32_t *alist, int32_t alias_cap, int64_t hash_slots, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(workspace && info && vsid && nbr && alist && n_cap > 0 && h_cap > 0 && h_cap <= h_cap_build && alias_cap > 0);
    const int64_t hcap = pick_hash_slots(n_cap, hash_slots);
    EFGH_CHECK_ARG(hcap > 0);
    const WsLayout w = ws_layout(n_cap, h_cap_build, nsamples, hcap);
    const char *ws = (const char *)workspace;
    int grid = cdiv((int64_t)h_cap * 16, TPB);
    if (grid > 8192) grid = 8192;
    k_neighbors<<<grid, TPB, 0, st>>>((const int4 *)(ws + w.vkeys), (const int *)(ws + w.mm),
                                      (const unsigned long long *)(ws + w.hkeys), (const int *)(ws + w.hvals), hcap - 1, info,
                                      h_cap, nbr, vsid, nsamples, (int2 *)alist, alias_cap);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// partitioned build: host side
namespace {

struct PartLayout {
    int64_t ent, toff, cumul, where, larr, cursor, wbase, biglist, gcount, grec, gvix, vrec, table, wprefix, bsum, mm, part, total;
    int W, nblk, ntiles, tp, ppt;
};

int ilog2(int64_t v) { int b = 0; while ((1LL << b) < v) ++b; return b; }

// tile of the scatter kernel: 8 points per thread when that still leaves >= 192 tiles, else 2
int part_ppt(int32_t n_cap) { return (int64_t)n_cap >= 192LL * STP * 8 ? 8 : 2; }

// entries per bucket: 2048 (four workgroups of k_lat_bucket per CU) unless the level needs more than NBMAX buckets of ~1000
int part_maxe(int32_t n_cap) { return (int64_t)n_cap * 4 > (int64_t)NBMAX * 1024 ? 4096 : 2048; }

PartLayout part_layout(int32_t n_cap, int32_t h_cap, int32_t nsamples, int nb, int S) {
    PartLayout w;
    int64_t o = 0;
    const int maxe = part_maxe(n_cap);
    w.ppt = part_ppt(n_cap);
    w.tp = STP * w.ppt;
    w.ntiles = cdiv(n_cap, w.tp);
    w.W = cdiv((int64_t)n_cap * 4, 32);
    w.nblk = cdiv(w.W, TPB * RWT);
    w.ent = o;     o += align256((int64_t)w.ntiles * w.tp * 4 * 8);
    w.toff = o;    o += align256((int64_t)w.ntiles * (nb + 1) * 4);
    w.cumul = o;   o += align256((int64_t)nb * w.ntiles * 4);
    w.where = o;   o += align256((int64_t)n_cap * 16);
    w.larr = o;    o += align256(((int64_t)nb * maxe + (int64_t)n_cap * 4) * 2);      // (regular windows + the overflow area)
    w.cursor = o;  o += align256((int64_t)nb * 4);
    w.wbase = o;   o += align256((int64_t)nb * 4);
    w.biglist = o; o += align256((int64_t)BIGCAP * 4);
    w.gcount = o;  o += align256((int64_t)nb * 4);
    w.grec = o;    o += align256((int64_t)nb * S * 16);
    w.gvix = o;    o += align256((int64_t)nb * S * 4);
    w.vrec = o;    o += align256((int64_t)h_cap * 4);
    w.table = o;   o += align256((int64_t)nb * S * 16);
    w.wprefix = o; o += align256((int64_t)w.W * 4);
    w.bsum = o;    o += align256((int64_t)w.nblk * 4);
    w.mm = o;      o += align256((int64_t)nsamples * 32);
    w.part = o;    o += align256(((int64_t)cdiv(n_cap, 64) + 4) * 48);
    w.total = o;
    return w;
}

bool part_args_ok(int32_t n_cap, int32_t nb, int32_t S) {
    return nb >= 2 && nb <= NBMAX && !(nb & (nb - 1)) && S >= 16 && S <= SMAX && !(S & (S - 1)) &&
           cdiv(n_cap, STP * part_ppt(n_cap)) <= NTMAX;
}

LatPart part_ptrs(char *ws, char *zeroed, const PartLayout &w, int32_t n_cap, int nb, int S, int want_off = 1, int use_big = 0) {
    LatPart P;
    P.want_off = want_off;
    P.use_big = use_big;
    P.ent = (unsigned long long *)(ws + w.ent);
    P.toff = (int *)(ws + w.toff);
    P.cumul = (int *)(ws + w.cumul);
    P.where = (unsigned *)(ws + w.where);
    P.larr = (unsigned short *)(ws + w.larr);
    P.cursor = (int *)(ws + w.cursor);
    P.wbase = (int *)(ws + w.wbase);
    P.biglist = (int *)(ws + w.biglist);
    P.big = (int *)zeroed + 1;
    P.gcount = (int *)(ws + w.gcount);
    P.grec = (int4 *)(ws + w.grec);
    P.gvix = (int *)(ws + w.gvix);
    P.vrec = (int *)(ws + w.vrec);
    P.table = (int4 *)(ws + w.table);
    P.ticket = (int *)zeroed;
    P.fbits = (unsigned *)(zeroed + 256);
    P.wprefix = (int *)(ws + w.wprefix);
    P.bsum = (int *)(ws + w.bsum);
    P.nb = nb; P.bb = ilog2(nb); P.S = S; P.sb = ilog2(S); P.maxe = part_maxe(n_cap);
    P.ntiles = w.ntiles; P.tp = w.tp; P.fb = ilog2((int64_t)n_cap * 4);
    return P;
}

float part_std32() {
    const uint32_t std_bits = 0x405105ECu;         // float32(4*sqrt(2/3)), generate_data.py:19
    float std32;
    memcpy(&std32, &std_bits, 4);
    return std32;
}

}  // namespace

extern "C" int32_t efgh_lattice_part_max_entries(int32_t n_cap) { return part_maxe(n_cap); }
/* elements of `list`: one window of max_entries per bucket + an overflow area for the buckets that hold more */
extern "C" int64_t efgh_lattice_part_list_len(int32_t n_cap, int32_t nbuckets) {
    return (int64_t)nbuckets * part_maxe(n_cap) + (int64_t)n_cap * 4;
}

extern "C" int32_t efgh_lattice_part_buckets(int32_t n_cap) {
    // ~1000 entries per bucket on average (half of the 2048 a bucket holds: room for the spread of the bucket sizes), ~2000 of
    // 4096 when that would take more than NBMAX buckets; 0 = too many points even so (the hash build serves those)
    const int64_t per = part_maxe(n_cap) == 2048 ? 1024 : 2048;
    int64_t nb = 8;
    while (nb * per < (int64_t)n_cap * 4) nb <<= 1;
    return (nb > NBMAX || cdiv(n_cap, STP * part_ppt(n_cap)) > NTMAX) ? 0 : (int32_t)nb;
}

extern "C" int64_t efgh_lattice_part_workspace_bytes(int32_t n_cap, int32_t h_cap, int32_t nsamples, int32_t nbuckets,
                                                     int32_t slots) {
    if (!part_args_ok(n_cap, nbuckets, slots)) return -1;
    return part_layout(n_cap, h_cap, nsamples, nbuckets, slots).total;
}

extern "C" int64_t efgh_lattice_part_zeroed_bytes(int32_t n_cap) {
    return 256 + align256((int64_t)cdiv((int64_t)n_cap * 4, 32) * 4);
}

extern "C" int efgh_lattice_part_build(const float *pts, int64_t pts_cstride, const int32_t *n_dev, int32_t n_cap,
                                       const int32_t *sid, int32_t pts_per_sample, int32_t nsamples, float scale32,
                                       float *bary, float *emg, int32_t *list, int32_t h_cap, int32_t *info, void *workspace,
                                       void *zeroed, int32_t nbuckets, int32_t slots, int32_t want_off, int32_t big_buckets,
                                       void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(n_cap > 0 && n_cap < (1 << 27) && h_cap > 0 && nsamples >= 1 && nsamples <= EFGH_LATTICE_MAX_SAMPLES);
    EFGH_CHECK_ARG(sid || pts_per_sample > 0);
    EFGH_CHECK_ARG(pts && bary && emg && list && info && workspace && zeroed);
    EFGH_CHECK_ARG(part_args_ok(n_cap, nbuckets, slots));
    const PartLayout w = part_layout(n_cap, h_cap, nsamples, nbuckets, slots);
    char *ws = (char *)workspace;
    const LatPart P = part_ptrs(ws, (char *)zeroed, w, n_cap, nbuckets, slots, want_off ? 1 : 0, big_buckets ? 1 : 0);
    int *mm = (int *)(ws + w.mm), *part = (int *)(ws + w.part);
    const float std32 = part_std32();
    const int nbp = cdiv(n_cap, TPB);
    const int pps = sid ? 1 : pts_per_sample;
    k_lat_keys<<<nbp, TPB, 0, st>>>(pts, pts_cstride, n_dev, n_cap, scale32, std32, (float4 *)bary, (float4 *)emg, part, sid, pps);
    k_lat_minmax<<<1, MMT, 0, st>>>(part, pts, pts_cstride, n_dev, n_cap, scale32, std32, sid, pps, nsamples, mm);
    if (w.ppt == 8)
        k_lat_scatter<8><<<w.ntiles, STP, 0, st>>>(pts, pts_cstride, n_dev, n_cap, scale32, std32, mm, sid, pps, nsamples, P, info);
    else
        k_lat_scatter<2><<<w.ntiles, STP, 0, st>>>(pts, pts_cstride, n_dev, n_cap, scale32, std32, mm, sid, pps, nsamples, P, info);
    if (P.maxe == 2048) {           // (512 threads per bucket measured the same as 256: 259 vs 261 us at level 0)
        const size_t lds = (size_t)slots * 20 + (size_t)2048 * 14;          // 68 KB with slots = 2048: above the 64 KB a launch gets by default
        static std::atomic<unsigned long long> raised_small{0};
        if (lds > 64 * 1024 && !efgh_raise_lds_once(raised_small, (const void *)k_lat_bucket<2048, 256>, 20 * SMAX + 14 * 2048)) {
            efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_lat_bucket", __FILE__, __LINE__);
            return EFGH_E_LAUNCH;
        }
        k_lat_bucket<2048, 256><<<nbuckets, 256, lds, st>>>(P, n_dev, n_cap, list, info);
    } else {
        const size_t lds = (size_t)slots * 20 + (size_t)4096 * 14;          // up to 98 KB: above the 64 KB a launch gets by default
        static std::atomic<unsigned long long> raised{0};
        if (!efgh_raise_lds_once(raised, (const void *)k_lat_bucket<4096, 512>, 20 * SMAX + 14 * 4096)) {
            efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_lat_bucket", __FILE__, __LINE__);
            return EFGH_E_LAUNCH;
        }
        k_lat_bucket<4096, 512><<<nbuckets, 512, lds, st>>>(P, n_dev, n_cap, list, info);
    }
    if (big_buckets) {
        const size_t lds = (size_t)slots * 20 + (size_t)MAXE_BIG * 14;      // 121-155 KB: above the 64 KB a launch gets by default
        static std::atomic<unsigned long long> raised_big{0};
        if (!efgh_raise_lds_once(raised_big, (const void *)k_lat_bucket_big, 20 * SMAX + 14 * MAXE_BIG)) {
            efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_lat_bucket_big", __FILE__, __LINE__);
            return EFGH_E_LAUNCH;
        }
        k_lat_bucket_big<<<256, 1024, lds, st>>>(P, n_dev, n_cap, list, info);         // (one 152-KB workgroup per CU)
    }
    k_lat_rank<<<w.nblk, TPB, 0, st>>>(P, w.W, info, h_cap);
    k_lat_number<<<cdiv((int64_t)nbuckets * slots, TPB), TPB, 0, st>>>(P, h_cap);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

extern "C" int efgh_lattice_part_neighbors(const void *workspace, const float *pts, int64_t pts_cstride, const int32_t *n_dev,
                                           int32_t n_cap, const int32_t *sid, int32_t pts_per_sample, int32_t nsamples,
                                           float scale32, float div32, int32_t h_cap_build, int32_t *info, int32_t h_cap,
                                           int32_t *nbr, int32_t *alist, int32_t alias_cap, int32_t *off, int32_t *vseg,
                                           float *pts_next, int32_t *vsid, int32_t nbuckets, int32_t slots, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(workspace && pts && info && vsid && nbr && alist && vseg && pts_next && n_cap > 0 && h_cap > 0 &&
                   h_cap <= h_cap_build && alias_cap > 0 && (sid || pts_per_sample > 0));
    EFGH_CHECK_ARG(part_args_ok(n_cap, nbuckets, slots));
    const PartLayout w = part_layout(n_cap, h_cap_build, nsamples, nbuckets, slots);
    char *ws = (char *)workspace;
    const LatPart P = part_ptrs(ws, ws, w, n_cap, nbuckets, slots);        // (the zeroed area is not used any more)
    LatGeom G;
    G.pts = pts; G.cstride = pts_cstride; G.scale32 = scale32; G.std32 = part_std32(); G.div32 = div32; G.sid = sid;
    G.pps = sid ? 1 : pts_per_sample;
    int gn = cdiv((int64_t)h_cap * 16, TPB);
    if (gn > 8192) gn = 8192;
    const int go = off ? cdiv(n_cap, TPB) : 0;       // (off == NULL: the build was told want_off = 0)
    int gv = cdiv(h_cap, TPB);
    if (gv > 2048) gv = 2048;
    k_lat_nbr<<<gn + go + gv, TPB, 0, st>>>(P, G, (const int *)(ws + w.mm), info, h_cap, nbr, nsamples, (int2 *)alist, alias_cap, gn, go,
                                            n_dev, n_cap, (int4 *)off, (int2 *)vseg, pts_next, h_cap_build, vsid);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}

/* ---- the tail of the pyramid in one launch (k_lat_tail): see include/efgh_hip.h efgh_lattice_tail_build ---- */
extern "C" int64_t efgh_lattice_tail_lds_bytes(int32_t slots) {
    return (int64_t)slots * 8 + (int64_t)slots * 4 * 6 + 16 + 1024 * 4 + (8 + 16 + 4) * 4 + (int64_t)4 * TAIL_NMAX * 2 + 64;
}

extern "C" int32_t efgh_lattice_tail_max_points(void) { return TAIL_NMAX; }

extern "C" int efgh_lattice_tail_build(const efgh_lattice_tail_desc *d, void *stream_) {
    hipStream_t st = (hipStream_t)stream_;
    EFGH_CHECK_ARG(d && d->nlevels >= 1 && d->nlevels <= TAIL_MAX_LEVELS && d->nsamples >= 1 && d->nsamples <= 64 && d->pts);
    EFGH_CHECK_ARG(d->slots >= 1024 && d->slots <= 2048 && (d->slots & (d->slots - 1)) == 0);
    EFGH_CHECK_ARG(d->info_prev ? d->prev_h_cap > 0 : (d->pts_per_sample > 0 && d->pts_per_sample <= TAIL_NMAX));
    TailArgs a;
    memset(&a, 0, sizeof(a));
    a.nlevels = d->nlevels; a.nsamples = d->nsamples; a.S = d->slots; a.pps = d->pts_per_sample;
    a.pts = d->pts; a.cstride = d->pts_cstride; a.info_prev = d->info_prev; a.prev_h_cap = d->prev_h_cap; a.std32 = part_std32();
    for (int l = 0; l < d->nlevels; ++l) {
        const efgh_lattice_tail_level &s = d->levels[l];
        EFGH_CHECK_ARG(s.bary && s.emg && s.list && s.vseg && s.nbr && s.pts_next && s.vsid && s.info && s.alist && s.h_cap > 0 && s.alias_cap > 0);
        TailLevel &t = a.lv[l];
        t.scale32 = s.scale32; t.div32 = s.div32; t.h_cap = s.h_cap; t.alias_cap = s.alias_cap;
        t.bary = (float4 *)s.bary; t.emg = (float4 *)s.emg; t.off = (int4 *)s.off; t.list = s.list; t.vseg = (int2 *)s.vseg; t.nbr = s.nbr;
        t.pts_next = s.pts_next; t.vsid = s.vsid; t.info = s.info; t.alist = (int2 *)s.alist;
    }
    const size_t lds = (size_t)efgh_lattice_tail_lds_bytes(d->slots);
    static std::atomic<unsigned long long> raised{0};
    if (!efgh_raise_lds_once(raised, (const void *)k_lat_tail, 160 * 1024)) {
        efgh_set_error("%s:%d: cannot raise the dynamic LDS limit of k_lat_tail", __FILE__, __LINE__);
        return EFGH_E_LAUNCH;
    }
    k_lat_tail<<<d->nsamples, TAIL_THREADS, lds, st>>>(a);
    EFGH_CHECK_LAUNCH();
    return EFGH_OK;
}
