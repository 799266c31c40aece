// Probe: random integer atomics on gfx950 - device ("agent") scope vs XCD-local (L2) scope.
//   hipcc --offload-arch=gfx950 -O3 -o probe_atomics probe_atomics.hip && ./probe_atomics
// Question behind it (BCL index pipeline): the lattice hash insert / CSR count are ~4 M random 4-8 B atomics per
// level; at agent scope they execute at the memory side (~20-28 G/s chip-wide measured in round 1).  If every workgroup that
// touches one table runs on ONE XCD (XCC_ID read at run time), the atomics can stay in that XCD's L2 (workgroup scope = no
// sc1 bit).  This probe measures both rates and checks that XCD-local atomics lose no update.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}

// MODE 0: agent-scope add32   1: workgroup-scope add32   2: agent min64   3: workgroup min64  4: agent CAS64  5: wg CAS64
template <int MODE>
__global__ void __launch_bounds__(256) k_flat(unsigned long long *t64, unsigned *t32, uint64_t mask, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        uint64_t h = mix64((uint64_t)i) & mask;
        if (MODE == 0) __hip_atomic_fetch_add(&t32[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) __hip_atomic_fetch_add(&t32[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_min(&t64[h], (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) __hip_atomic_fetch_min(&t64[h], (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 4) { unsigned long long e = ~0ULL; __hip_atomic_compare_exchange_strong(&t64[h], &e, (unsigned long long)i, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (MODE == 5) { unsigned long long e = ~0ULL; __hip_atomic_compare_exchange_strong(&t64[h], &e, (unsigned long long)i, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
}

// XCD-affine: NP partitions, partition p is owned by the first XCD that claims it; every workgroup works only on partitions
// its own XCD owns (chunks handed out by a per-partition cursor).  Item i belongs to partition mix(i) % NP.  All atomics on the
// partition's table are workgroup scope (stay in the owning XCD's L2).  The items of a partition are found by scanning ALL items
// (as the lattice kernels would scan all entries of a sample).
constexpr int NP = 8;
constexpr int CHUNK = 4096;
struct Ctl { int owner[NP]; int cursor[NP]; int done[NP]; };

__global__ void __launch_bounds__(256) k_xcd(unsigned *t32, unsigned long long *t64, uint64_t mask_part, long long n, Ctl *ctl) {
    __shared__ int s_p, s_chunk;
    const int me = xcc_id();
    const int nchunks = (int)((n + CHUNK - 1) / CHUNK);
    for (;;) {
        if (threadIdx.x == 0) {
            int p = -1, c = -1;
            for (int q = 0; q < NP && p < 0; ++q) {         // a partition my XCD owns that still has chunks
                if (__hip_atomic_load(&ctl->owner[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == me) {
                    c = atomicAdd(&ctl->cursor[q], 1);
                    if (c < nchunks) p = q;
                }
            }
            for (int q = 0; q < NP && p < 0; ++q) {         // else claim an unowned one
                if (atomicCAS(&ctl->owner[q], -1, me) == -1) { c = atomicAdd(&ctl->cursor[q], 1); if (c < nchunks) p = q; }
            }
            s_p = p; s_chunk = c;
        }
        __syncthreads();
        const int p = s_p, c = s_chunk;
        __syncthreads();
        if (p < 0) return;
        unsigned *tp32 = t32 + (size_t)p * (mask_part + 1);
        unsigned long long *tp64 = t64 + (size_t)p * (mask_part + 1);
        const long long i0 = (long long)c * CHUNK;
        for (int j = threadIdx.x; j < CHUNK; j += 256) {
            long long i = i0 + j;
            if (i >= n) break;
            uint64_t m = mix64((uint64_t)i);
            if ((int)(m % NP) != p) continue;
            uint64_t h = (m >> 8) & mask_part;
            __hip_atomic_fetch_add(&tp32[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(&tp64[h], (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// LDS form: workgroup (b, q) owns partition q of sample b: scans all of the sample's items (8-byte packed keys read from
// global memory), keeps the ones of its partition, and does ds_min_u64 + ds_add_u32 on a table in LDS.
template <int QP>
__global__ void __launch_bounds__(1024) k_lds(const unsigned long long *__restrict__ keys, long long n_per, unsigned *out) {
    __shared__ unsigned long long tk[8192];
    __shared__ unsigned tc[8192];
    const int b = blockIdx.x / QP, q = blockIdx.x % QP;
    for (int i = threadIdx.x; i < 8192; i += 1024) { tk[i] = ~0ULL; tc[i] = 0; }
    __syncthreads();
    const unsigned long long *kp = keys + (long long)b * n_per;
    for (long long i = threadIdx.x; i < n_per; i += 1024) {
        unsigned long long k = kp[i];
        uint64_t m = mix64(k);
        if ((int)(m % QP) != q) continue;
        unsigned h = (unsigned)(m >> 8) & 8191u;
        atomicMin(&tk[h], (unsigned long long)i);
        atomicAdd(&tc[h], 1u);
    }
    __syncthreads();
    unsigned s = 0;
    for (int i = threadIdx.x; i < 8192; i += 1024) s += tc[i] + (unsigned)tk[i];
    if (s == 0x12345678u) out[blockIdx.x] = s;
    if (threadIdx.x == 0) { unsigned t = 0; for (int i = 0; i < 8192; ++i) t += tc[i]; out[blockIdx.x] = t; }
}

__global__ void k_fillkeys(unsigned long long *k, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) k[i] = mix64((uint64_t)i * 7919u) % 31000;      // ~31 k distinct keys per sample
}

__global__ void k_census(int *cnt) { if (threadIdx.x == 0) atomicAdd(&cnt[xcc_id()], 1); }

int main() {
    const long long n = 4 << 20;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int *cen; CK(hipMalloc(&cen, 64)); CK(hipMemset(cen, 0, 64));
    k_census<<<2048, 64>>>(cen);
    int hc[16]; CK(hipMemcpy(hc, cen, 64, hipMemcpyDeviceToHost));
    printf("census of 2048 blocks over XCC_ID:"); for (int i = 0; i < 16; ++i) printf(" %d", hc[i]); printf("\n");
    for (int tl = 17; tl <= 23; tl += 3) {
        const uint64_t T = 1ull << tl;
        unsigned long long *t64; unsigned *t32;
        CK(hipMalloc(&t64, T * 8)); CK(hipMalloc(&t32, T * 4));
        const char *names[6] = {"add32 agent", "add32 workgroup", "min64 agent", "min64 workgroup", "cas64 agent", "cas64 workgroup"};
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e9f;
            unsigned long long sum = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemset(t64, 0xFF, T * 8)); CK(hipMemset(t32, 0, T * 4));
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: k_flat<0><<<4096, 256>>>(t64, t32, T - 1, n); break;
                    case 1: k_flat<1><<<4096, 256>>>(t64, t32, T - 1, n); break;
                    case 2: k_flat<2><<<4096, 256>>>(t64, t32, T - 1, n); break;
                    case 3: k_flat<3><<<4096, 256>>>(t64, t32, T - 1, n); break;
                    case 4: k_flat<4><<<4096, 256>>>(t64, t32, T - 1, n); break;
                    case 5: k_flat<5><<<4096, 256>>>(t64, t32, T - 1, n); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            if (mode < 2) {
                std::vector<unsigned> h(T); CK(hipMemcpy(h.data(), t32, T * 4, hipMemcpyDeviceToHost));
                for (auto v : h) sum += v;
            }
            printf("table 2^%d slots  %-16s %8.1f us  %6.1f G atomics/s   %s\n", tl, names[mode], best * 1e3, n / (best * 1e-3) / 1e9,
                   mode < 2 ? (sum == (unsigned long long)n ? "sum ok" : "SUM WRONG (lost updates)") : "");
        }
        CK(hipFree(t64)); CK(hipFree(t32));
    }
    // XCD-affine partitioned form: 8 partitions of 2^17 slots (0.5 MB u32 + 1 MB u64 each)
    for (int tl = 14; tl <= 20; tl += 3) {
        const uint64_t Tp = 1ull << tl;
        unsigned long long *t64; unsigned *t32; Ctl *ctl;
        CK(hipMalloc(&t64, NP * Tp * 8)); CK(hipMalloc(&t32, NP * Tp * 4)); CK(hipMalloc(&ctl, sizeof(Ctl)));
        float best = 1e9f; bool ok = true;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(t64, 0xFF, NP * Tp * 8)); CK(hipMemset(t32, 0, NP * Tp * 4));
            Ctl h; for (int q = 0; q < NP; ++q) { h.owner[q] = -1; h.cursor[q] = 0; h.done[q] = 0; }
            CK(hipMemcpy(ctl, &h, sizeof(Ctl), hipMemcpyHostToDevice));
            CK(hipEventRecord(e0));
            k_xcd<<<2048, 256>>>(t32, t64, Tp - 1, n, ctl);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            std::vector<unsigned> hv(NP * Tp); CK(hipMemcpy(hv.data(), t32, NP * Tp * 4, hipMemcpyDeviceToHost));
            unsigned long long sum = 0; for (auto v : hv) sum += v;
            if (sum != (unsigned long long)n) { ok = false; printf("  rep %d: sum %llu != %lld\n", rep, sum, n); }
            CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            if (rep == 0) { printf("  owners:"); for (int q = 0; q < NP; ++q) printf(" %d", h.owner[q]); printf("\n"); }
        }
        printf("XCD-affine, 8 partitions x 2^%d slots (add32 + min64 per item, 8x scan): %8.1f us  %6.1f G items/s  %s\n", tl,
               best * 1e3, n / (best * 1e-3) / 1e9, ok ? "sums ok" : "LOST UPDATES");
        CK(hipFree(t64)); CK(hipFree(t32)); CK(hipFree(ctl));
    }
    {
        const long long n_per = 512 << 10;
        unsigned long long *keys; unsigned *out;
        CK(hipMalloc(&keys, 8 * n_per * 8)); CK(hipMalloc(&out, 4096 * 4));
        k_fillkeys<<<(unsigned)((8 * n_per + 255) / 256), 256>>>(keys, 8 * n_per);
        for (int qp = 8; qp <= 64; qp *= 2) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (qp == 8) k_lds<8><<<8 * 8, 1024>>>(keys, n_per, out);
                if (qp == 16) k_lds<16><<<8 * 16, 1024>>>(keys, n_per, out);
                if (qp == 32) k_lds<32><<<8 * 32, 1024>>>(keys, n_per, out);
                if (qp == 64) k_lds<64><<<8 * 64, 1024>>>(keys, n_per, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<unsigned> h(8 * qp); CK(hipMemcpy(h.data(), out, 8 * qp * 4, hipMemcpyDeviceToHost));
            unsigned long long sum = 0; for (auto v : h) sum += v;
            printf("LDS partition hash: 8 samples x %d partitions (1024-thread WGs), 512k items/sample: %8.1f us  %s\n", qp, best * 1e3,
                   sum == 8ull * n_per ? "sum ok" : "SUM WRONG");
        }
    }
    return 0;
}
