/* TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the reference's permutohedral-lattice
 * builder.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product (efgh_amd/) never does.
 *
 * Restates, in plain scalar C:
 *   GenerateData.get_keys_and_barycentric      nets/generate_data.py:56-112
 *   GenerateData.__call__ (5-level loop)       nets/generate_data.py:117-193
 *   key2int                                    nets/transforms.py:62-77
 *   build_it                                   nets/transforms.py:125-184
 *   Traverse offsets (radius 1, d=3)           nets/transforms.py:104-122  (values: lattice_consts.npz)
 * The int64->int64 map of lib/khash.h is replaced by a private open-addressing table: only the
 * get/set results are observable on this path.
 *
 * Parity pinning: tests/test_oracle_lattice.py checks this file bit-for-bit against
 * tests/golden/lattice_n{512,4096}.npz and the sha256 known answers in lattice_kat.json, all
 * produced by the unmodified reference python (tests/golden/make_golden.py).
 *
 * Float recipe (bit-exact vs torch CPU / MKL): elevate is an FMA chain in
 * column order, scalars are rounded to fp32 first.  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define D1 4
#define NB 15
#define MAXL 8

/* elevate_mat (generate_data.py:15-20) as fp32 bit patterns taken from the reference. */
static const uint32_t ELEV_BITS[4][3] = {
    {0x3F3504F3u, 0x3ED105EBu, 0x3E93CD3Au},
    {0xBF3504F3u, 0x3ED105EBu, 0x3E93CD3Au},
    {0x00000000u, 0xBF5105EBu, 0x3E93CD3Au},
    {0x00000000u, 0x00000000u, 0xBF5DB3D7u}};
/* canonical simplex (generate_data.py:26-30) */
static const int CANON[4][4] = {{0, 1, 2, 3}, {0, 1, 2, -1}, {0, 1, -2, -1}, {0, -3, -2, -1}};
/* radius-1 neighbour offsets in Traverse order (generate_data.py:44-52) */
static const int NBR_OFF[NB][4] = {
    {0, 0, 0, 0},   {-1, -1, -1, 3}, {-1, -1, 3, -1}, {-2, -2, 2, 2},  {-1, 3, -1, -1},
    {-2, 2, -2, 2}, {-2, 2, 2, -2},  {-3, 1, 1, 1},   {3, -1, -1, -1}, {2, -2, -2, 2},
    {2, -2, 2, -2}, {1, -3, 1, 1},   {2, 2, -2, -2},  {1, 1, -3, 1},   {1, 1, 1, -3}};

static float f_from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

/* ---------------------------------------------------------------- per-point keys ---------- */
/* generate_data.py:56-112 for one point. pos = already-scaled (x,y,z). */
static void point_keys(const float pos[3], float std32, int64_t key[4][4] /*[coord][rem]*/,
                       float bary[4], float emg_out[4]) {
    float el[4], gr[4], emg[4];
    int rank[4];
    for (int r = 0; r < 4; ++r) {
        float e0 = f_from_bits(ELEV_BITS[r][0]), e1 = f_from_bits(ELEV_BITS[r][1]),
              e2 = f_from_bits(ELEV_BITS[r][2]);
        float acc = e0 * pos[0];                 /* :67  matmul as fma chain */
        acc = fmaf(e1, pos[1], acc);
        acc = fmaf(e2, pos[2], acc);
        el[r] = acc * std32;
        gr[r] = rintf(el[r] / 4.0f) * 4.0f;      /* :70  round-half-even */
        emg[r] = el[r] - gr[r];                  /* :72 */
    }
    for (int r = 0; r < 4; ++r) {                /* :73-78 descending sort position, stable */
        int c = 0;
        for (int j = 0; j < 4; ++j)
            if (emg[j] > emg[r] || (emg[j] == emg[r] && j < r)) ++c;
        rank[r] = c;
    }
    float rs = (gr[0] + gr[1] + gr[2] + gr[3]) / 4.0f;   /* :80 */
    for (int r = 0; r < 4; ++r) {                /* :82-92 */
        float rf = (float)rank[r];
        int cond = ((rf >= 4.0f - rs) && (rs > 0.0f)) || ((rf < -rs) && (rs < 0.0f));
        float sign = (rs > 0.0f) ? -1.0f : ((rs < 0.0f) ? 1.0f : 0.0f);
        float adj = 4.0f * sign * (cond ? 1.0f : 0.0f);
        gr[r] += adj;
        rank[r] += (int)adj;
        rank[r] += (int)rs;
    }
    float b5[5] = {0, 0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) emg[r] = el[r] - gr[r];          /* :95 */
    for (int r = 0; r < 4; ++r) b5[3 - rank[r]] += emg[r];      /* :99 */
    for (int r = 0; r < 4; ++r) b5[4 - rank[r]] -= emg[r];      /* :100 */
    for (int j = 0; j < 5; ++j) b5[j] /= 4.0f;                   /* :101 */
    b5[0] += 1.0f + b5[4];                                       /* :102 */
    for (int j = 0; j < 4; ++j) { bary[j] = b5[j]; emg_out[j] = emg[j]; }
    for (int c = 0; c < 4; ++c)
        for (int rem = 0; rem < 4; ++rem)
            key[c][rem] = (int64_t)gr[c] + CANON[rank[c]][rem]; /* :106 */
}

/* ---------------------------------------------------------------- int64 map ---------------- */
typedef struct { int64_t *k; int64_t *v; uint8_t *used; uint64_t cap; } map_t;
static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static void map_init(map_t *m, uint64_t n) {
    uint64_t cap = 64; while (cap < 2 * n + 16) cap <<= 1;
    m->cap = cap; m->k = malloc(cap * 8); m->v = malloc(cap * 8); m->used = calloc(cap, 1);
}
static void map_free(map_t *m) { free(m->k); free(m->v); free(m->used); }
static int64_t map_get(const map_t *m, int64_t key, int64_t dflt) {
    uint64_t i = mix64((uint64_t)key) & (m->cap - 1);
    while (m->used[i]) { if (m->k[i] == key) return m->v[i]; i = (i + 1) & (m->cap - 1); }
    return dflt;
}
static void map_set(map_t *m, int64_t key, int64_t val) {
    uint64_t i = mix64((uint64_t)key) & (m->cap - 1);
    while (m->used[i]) { if (m->k[i] == key) { m->v[i] = val; return; } i = (i + 1) & (m->cap - 1); }
    m->used[i] = 1; m->k[i] = key; m->v[i] = val;
}

/* transforms.py:62-77 (no range check: out-of-range neighbour keys alias, on purpose) */
static int64_t key2int(const int64_t key[4], const int64_t mins[4], const int64_t maxs[4]) {
    int64_t res = 0;
    for (int i = 0; i < 3; ++i) {
        res += key[i] - mins[i];
        res *= (maxs[i + 1] - mins[i + 1] + 1);
    }
    res += key[3] - mins[3];
    return res;
}

/* ---------------------------------------------------------------- whole pyramid ------------ */
typedef struct {
    int L;
    int64_t N[MAXL], H[MAXL];
    float *bary[MAXL], *emg[MAXL];       /* (4,N) row-major */
    int64_t *off[MAXL];                  /* (4,N) */
    int64_t *nbr[MAXL];                  /* (15,H) */
    float *pts_next[MAXL];               /* (3,H): points handed to the next level */
    int64_t mins[MAXL][4], maxs[MAXL][4];
} gd_t;

void *oracle_gd_run(const float *pc /* (3,N) */, int64_t N, int L, const double *scales) {
    gd_t *g = calloc(1, sizeof(gd_t));
    g->L = L;
    const double expected_std = 4.0 * sqrt(2.0 / 3.0);           /* generate_data.py:19 */
    const float std32 = (float)expected_std;
    float *cur = malloc(sizeof(float) * 3 * (size_t)N);          /* last_pc1 */
    memcpy(cur, pc, sizeof(float) * 3 * (size_t)N);
    int64_t n = N;
    for (int l = 0; l < L; ++l) {
        const float s32 = (float)scales[l];
        for (int64_t i = 0; i < 3 * n; ++i) cur[i] *= s32;       /* :130 */
        int64_t *keys = malloc(sizeof(int64_t) * 16 * (size_t)n); /* [p][rem][coord] */
        g->N[l] = n;
        g->bary[l] = malloc(sizeof(float) * 4 * (size_t)n);
        g->emg[l] = malloc(sizeof(float) * 4 * (size_t)n);
        g->off[l] = malloc(sizeof(int64_t) * 4 * (size_t)n);
        int64_t mins[4], maxs[4];
        for (int c = 0; c < 4; ++c) { mins[c] = INT64_MAX; maxs[c] = INT64_MIN; }
        for (int64_t p = 0; p < n; ++p) {
            float pos[3] = {cur[p], cur[n + p], cur[2 * n + p]};
            int64_t k[4][4]; float b[4], e[4];
            point_keys(pos, std32, k, b, e);
            for (int r = 0; r < 4; ++r) { g->bary[l][r * n + p] = b[r]; g->emg[l][r * n + p] = e[r]; }
            for (int rem = 0; rem < 4; ++rem)
                for (int c = 0; c < 4; ++c) {
                    keys[(p * 4 + rem) * 4 + c] = k[c][rem];
                    if (k[c][rem] < mins[c]) mins[c] = k[c][rem];   /* :135-136 */
                    if (k[c][rem] > maxs[c]) maxs[c] = k[c][rem];
                }
        }
        memcpy(g->mins[l], mins, sizeof(mins)); memcpy(g->maxs[l], maxs, sizeof(maxs));
        /* build_it part (i): first-seen numbering, transforms.py:153-166 */
        map_t m; map_init(&m, (uint64_t)(4 * n));
        int64_t *vkey = malloc(sizeof(int64_t) * 16 * (size_t)n);   /* vertex keys, [h][coord] */
        int64_t cnt = 0;
        for (int64_t p = 0; p < n; ++p)
            for (int rem = 0; rem < 4; ++rem) {
                const int64_t *kk = &keys[(p * 4 + rem) * 4];
                int64_t ki = key2int(kk, mins, maxs);
                int64_t idx = map_get(&m, ki, -1);
                if (idx == -1) {
                    map_set(&m, ki, cnt);
                    memcpy(&vkey[cnt * 4], kk, 4 * sizeof(int64_t));
                    idx = cnt++;
                }
                g->off[l][rem * n + p] = idx;
            }
        g->H[l] = cnt;
        /* part (ii): neighbours, transforms.py:168-180 */
        g->nbr[l] = malloc(sizeof(int64_t) * NB * (size_t)cnt);
        for (int64_t h = 0; h < cnt; ++h)
            for (int t = 0; t < NB; ++t) {
                int64_t nk[4];
                for (int c = 0; c < 4; ++c) nk[c] = vkey[h * 4 + c] + NBR_OFF[t][c];
                g->nbr[l][t * cnt + h] = map_get(&m, key2int(nk, mins, maxs), -1);
            }
        /* next-level points: generate_data.py:176-178 (keys stored as fp32, :163) */
        float *nxt = malloc(sizeof(float) * 3 * (size_t)(cnt > 0 ? cnt : 1));
        const float div32 = (float)(expected_std * scales[l]);
        for (int64_t h = 0; h < cnt; ++h) {
            float kf[4];
            for (int c = 0; c < 4; ++c) kf[c] = (float)vkey[h * 4 + c] / div32;
            for (int j = 0; j < 3; ++j) {
                float acc = f_from_bits(ELEV_BITS[0][j]) * kf[0];
                acc = fmaf(f_from_bits(ELEV_BITS[1][j]), kf[1], acc);
                acc = fmaf(f_from_bits(ELEV_BITS[2][j]), kf[2], acc);
                acc = fmaf(f_from_bits(ELEV_BITS[3][j]), kf[3], acc);
                nxt[j * cnt + h] = acc;
            }
        }
        g->pts_next[l] = nxt;
        map_free(&m); free(vkey); free(keys); free(cur);
        cur = malloc(sizeof(float) * 3 * (size_t)(cnt > 0 ? cnt : 1));
        memcpy(cur, nxt, sizeof(float) * 3 * (size_t)cnt);
        n = cnt;
    }
    free(cur);
    return g;
}

int64_t oracle_gd_N(void *h, int l) { return ((gd_t *)h)->N[l]; }
int64_t oracle_gd_H(void *h, int l) { return ((gd_t *)h)->H[l]; }
/* what: 0 bary f32(4,N) 1 emg f32(4,N) 2 off i64(4,N) 3 nbr i64(15,H) 4 next pts f32(3,H)
 *       5 mins i64(4) 6 maxs i64(4) */
void oracle_gd_copy(void *h, int l, int what, void *dst) {
    gd_t *g = h; size_t n = (size_t)g->N[l], H = (size_t)g->H[l];
    switch (what) {
    case 0: memcpy(dst, g->bary[l], 4 * n * sizeof(float)); break;
    case 1: memcpy(dst, g->emg[l], 4 * n * sizeof(float)); break;
    case 2: memcpy(dst, g->off[l], 4 * n * sizeof(int64_t)); break;
    case 3: memcpy(dst, g->nbr[l], NB * H * sizeof(int64_t)); break;
    case 4: memcpy(dst, g->pts_next[l], 3 * H * sizeof(float)); break;
    case 5: memcpy(dst, g->mins[l], 4 * sizeof(int64_t)); break;
    case 6: memcpy(dst, g->maxs[l], 4 * sizeof(int64_t)); break;
    }
}
void oracle_gd_free(void *h) {
    gd_t *g = h;
    for (int l = 0; l < g->L; ++l) {
        free(g->bary[l]); free(g->emg[l]); free(g->off[l]); free(g->nbr[l]); free(g->pts_next[l]);
    }
    free(g);
}
