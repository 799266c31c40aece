/* libefgh_hip.so — C-ABI of the MI355X-native EFGHNet hot path.
 *
 * Conventions: every entry point returns int (0 = ok, <0 = EFGH_E_*);
 * `efgh_last_error()` returns a thread-local message.  Tensor arguments are raw DEVICE pointers
 * with explicit sizes/strides; dtypes are fixed per argument (float = fp32, int32_t indices).
 * The caller owns all memory including workspaces; nothing is allocated, freed or synchronised
 * inside; the last argument is the hipStream_t (as void*) the work is enqueued on.
 *
 * The reference has exactly one native interface (the cffi module `_khash_ffi`,
 * lib/khash_int2int.h:8-33, called from nets/transforms.py:149-183); everything else on the hot
 * path is python/torch.  Each entry point below cites the reference code it replaces.
 *
 * Layouts: image activations are channels-last  [B][H][W][C]  (C contiguous);
 * point/vertex features are row-major  [N][C];  the reference's (B,C,H,W)/(B,C,N) tensors are
 * converted at the model boundary only.
 */
#ifndef EFGH_HIP_H
#define EFGH_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFGH_OK 0
#define EFGH_E_INVALID (-1)   /* bad argument / unsupported shape */
#define EFGH_E_LAUNCH (-2)    /* hip launch error */
#define EFGH_WROTE_OUT 1      /* weight-gradient entry points only: the launch wrote `out`'s layout itself, dWp is untouched */

const char *efgh_last_error(void);

/* ABI version of this header; efgh_version() returns the one the library was built from.  A caller compiled against another
 * value must not call into the library (the Python binding refuses to load it).
 *   1  rounds 1-4.  NOTE: round 4 added the caller-owned `workspace` argument to efgh_thin_wgrad / efgh_c4_wgrad and a
 *      16-byte alignment requirement on their dWp without moving this number; out-of-tree callers of those two built
 *      against a round-3 header pass their stream as the workspace.
 *   2  round 5: that signature change is recorded here; efgh_plane_gemm / efgh_plane_wgrad (LDS-DMA staged batched plain GEMMs),
 *      the per-sample fused small-level lattice build, the split-precision entry points (efgh_gather_gemm_{bf16x3,bf16x6,f16x3},
 *      efgh_split_{bf16,f16}) removed.
 *      Added later in round 5 WITHOUT moving the number (no existing signature or struct changed; a caller built against the
 *      earlier version-2 header keeps working, a caller of the new entry points against an earlier library fails at symbol lookup):
 *      efgh_wino_pack_batched, efgh_pack_weight_batched_tiled, efgh_wino2d_output_pooled, efgh_c4_pooled_supported,
 *      efgh_c4_conv3x3_pooled, efgh_segment_workspace, efgh_segment_colmax_ws, efgh_segment_colmean_ws,
 *      efgh_wino_conv3x3_hpool, efgh_maxpool_v2, efgh_fold_unpack_arm, efgh_fold_unpack_disarm. */
/*   3  round 6: the weight-gradient entry points (efgh_gather_wgrad, efgh_thin_wgrad, efgh_c4n4_wgrad, efgh_c4_wgrad, efgh_sc_wgrad,
 *      efgh_wino_wgrad, efgh_wino2d_wfinish) take an explicit `const efgh_wgrad_out_desc *out` in front of the stream and return
 *      EFGH_WROTE_OUT when they left the gradient in the caller's layout themselves; efgh_fold_unpack_arm / _disarm (a thread-local
 *      descriptor consumed by the NEXT call: hidden state in an interface that promises none) are removed.  New:
 *      efgh_wino2d_input_act, efgh_wino2d_bwd_transforms(_pooled) (BatchNorm apply / backward apply inside the 2-D Winograd
 *      transforms); efgh_splat_gather / efgh_splat_bwd take `normalize` (args['bcn_use_norm'] of the reference's E net). */
#define EFGH_ABI_VERSION 3
int efgh_version(void);

/* ------------------------------------------------------------------ lattice (K1, K2) ------
 * replaces GenerateData.get_keys_and_barycentric  nets/generate_data.py:56-112,
 *          key2int / build_it                     nets/transforms.py:62-77, 125-184,
 *          the khash map                          lib/khash_int2int.h:8-33
 * for ONE pyramid level and ALL samples of a batch.  Bit-exact w.r.t. the reference for bary/emg (fp32 bits),
 * lattice_offset and blur_neighbors (integers).
 *
 * The points of `nsamples` frame-pairs are concatenated (sample-major); the sample of point p is sid[p], or
 * p / pts_per_sample when sid is NULL (level 0).  Every sample keeps its own key_mins/maxs and its own hash keys, vertices
 * are numbered sample-major, so (vertex index - first vertex of the sample) and the neighbour indices are exactly the
 * per-sample results of the reference.
 *
 * Data-dependent sizes live in DEVICE memory: the number of input points is *n_dev (NULL: n_cap is exact), capped by n_cap;
 * the number of vertices is written to info[EFGH_LATTICE_INFO_H] and must fit h_cap (else bit 0 of
 * info[EFGH_LATTICE_INFO_ERR] is set and nothing is written out of bounds).  Launch sizes derive from the capacities only,
 * so the levels of a pyramid can be enqueued back to back (level l+1: pts = pts_next, pts_cstride = h_cap, n_dev =
 * info + EFGH_LATTICE_INFO_H, n_cap = h_cap, sid = vsid) with one host read-back at the end.                              */
#define EFGH_LATTICE_INFO_H 0        /* number of vertices (pc1_hash_cnt summed over the samples) */
#define EFGH_LATTICE_INFO_ERR 1      /* bit 0: H > h_cap; bit 1: more aliased neighbour hits than alias_cap; bit 2: hash table / bucket full; bit 3 (with bit 2): key range too wide for the partitioned build's entry word */
#define EFGH_LATTICE_INFO_ALIAS 2    /* number of aliased neighbour hits (see efgh_lattice_level_neighbors) */
#define EFGH_LATTICE_INFO_CURSOR 3   /* internal (number of occupied hash slots == H) */
#define EFGH_LATTICE_INFO_SEG 4      /* info[SEG + b] = first vertex of sample b */
#define EFGH_LATTICE_MAX_SAMPLES 1024

/* entries of the level's default hash table (power of two >= 8*n_cap: 4*n_cap keys can never fill it).  `hash_slots` of the two
 * calls below selects a SMALLER table (power of two >= 4096; 0 = default) when the caller knows roughly how many vertices to
 * expect (e.g. from the previous batch): the build is bound by random probes into this table, and a table that fits the caches
 * is ~2x faster.  If the estimate was too small the table fills up, bit 2 of info[EFGH_LATTICE_INFO_ERR] is set, nothing is
 * written out of bounds, and the level has to be rebuilt with hash_slots = 0.  Both calls must get the same value.              */
int64_t efgh_lattice_hash_capacity(int32_t n_cap);
/* bytes of scratch for efgh_lattice_level_build; efgh_lattice_level_neighbors reads the same workspace afterwards */
int64_t efgh_lattice_workspace_bytes(int32_t n_cap, int32_t h_cap, int32_t nsamples);

/* pts: 3 coordinates, pts[c*pts_cstride + p]; positions are multiplied by `scale32` first
 *      (generate_data.py:130), `div32` = float32(expected_std*scale) (:177).
 * out, per point (point-major, 16 B per point and array):
 *      bary [n_cap][4] f32   (pc1_barycentric[r][p]  at [p][r])
 *      emg  [n_cap][4] f32   (pc1_el_minus_gr)
 *      off  [n_cap][4] i32   (pc1_lattice_offset, in [0,H))
 * out, per vertex (first-seen order):
 *      vseg [h_cap][2] i32   (start, length) of the vertex's list in `list`
 *      list [4*n_cap]  i32   the flat positions f = 4*p + r that splat onto the vertex, ascending (the inverse of `off`,
 *                            what efgh_splat_gather walks)
 *      pts_next[c*h_cap + h] f32   (next level's points, generate_data.py:176-178)
 *      vsid [h_cap]    i32   sample of vertex h (the next level's sid)
 *      info [EFGH_LATTICE_INFO_SEG + nsamples] i32                                              */
int efgh_lattice_level_build(const float *pts, int64_t pts_cstride, const int32_t *n_dev, int32_t n_cap,
                             const int32_t *sid, int32_t pts_per_sample, int32_t nsamples, float scale32,
                             float div32, float *bary, float *emg, int32_t *off, int32_t *list, int32_t h_cap,
                             int32_t *vseg, float *pts_next, int32_t *vsid, int32_t *info, void *workspace,
                             int64_t hash_slots, void *stream);

/* blur neighbours (transforms.py:168-180): nbr[h*16 + t] = index of vertex key(h)+offset_t, or -1, t < 15.
 * key2int has no range check (transforms.py:173-180): a neighbour key outside the sample's key box aliases to the integer
 * of another lattice point and may hit that vertex.  Such hits are reproduced and marked: bit t of nbr[h*16 + 15], one
 * record (h*16 + t, hit vertex) in alist [alias_cap][2], their number in info[EFGH_LATTICE_INFO_ALIAS]; all unmarked entries
 * form a symmetric relation (nbr[m][t] == h  <=>  nbr[h][15-t] == m), which efgh_table_gather_transposed relies on.
 * `workspace`, n_cap, h_cap_build, nsamples: as passed to efgh_lattice_level_build; h_cap (<= h_cap_build) rows of nbr.     */
int efgh_lattice_level_neighbors(const void *workspace, int32_t n_cap, int32_t h_cap_build, int32_t nsamples,
                                 int32_t *info, const int32_t *vsid, int32_t h_cap, int32_t *nbr, int32_t *alist,
                                 int32_t alias_cap, int64_t hash_slots, void *stream);

/* The TAIL of a pyramid in ONE launch (round 5): from the first level whose samples have at most efgh_lattice_tail_max_points()
 * points each, one workgroup per sample builds that level and every level below it entirely in LDS (the lattices of different
 * samples never interact; the level's vertices are the sample's points of the next level) - keys, extrema, an open-addressing
 * table of `slots` entries, first-seen numbering, vertex records, lattice_offset, ascending lists, the 15 neighbour probes -
 * and the workgroups meet once per level on a ticket to exchange the sample-major vertex bases.  Same arrays and meaning as
 * efgh_lattice_level_build / _neighbors per level, with the list of sample b's vertices inside [4 * first point of b, ...).
 * Input of the first tail level: the level above (pts = its pts_next, pts_cstride = prev_h_cap = its h_cap, info_prev = its
 * info: vertex count and per-sample bases), or - when the tail starts at level 0 - the cloud itself (info_prev NULL,
 * pts_per_sample points per sample).  Every level's info block must be zeroed and have EFGH_LATTICE_INFO_SEG + 2 * nsamples + 1
 * ints (the ticket and the per-sample counts live behind the bases).  A sample that does not fit (points, 0.8 * slots
 * vertices with slots <= 2 048, a list of more than 2 048 entries) sets bit 2 of that level's ERR word: rebuild with the per-level entry points.
 * nsamples <= 64 (the workgroups of the launch wait for each other).                                                        */
typedef struct efgh_lattice_tail_level {
    float scale32, div32;              /* as efgh_lattice_level_build */
    int32_t h_cap, alias_cap;
    float *bary, *emg; int32_t *off;   /* off may be NULL (inference) */
    int32_t *list, *vseg, *nbr; float *pts_next; int32_t *vsid, *info, *alist;
} efgh_lattice_tail_level;
typedef struct efgh_lattice_tail_desc {
    int32_t nlevels, nsamples, slots, pts_per_sample;
    const float *pts; int64_t pts_cstride; const int32_t *info_prev; int32_t prev_h_cap, pad;
    efgh_lattice_tail_level levels[5];
} efgh_lattice_tail_desc;
int32_t efgh_lattice_tail_max_points(void);
int64_t efgh_lattice_tail_lds_bytes(int32_t slots);
int efgh_lattice_tail_build(const efgh_lattice_tail_desc *d, void *stream);

/* The same level WITHOUT a global hash insert (round 3; the default of efgh_amd/lattice.py): every (point, corner) entry is
 * bucket-sorted inside its tile of points (bucket = top bits of a bijective mix of its key integer), one workgroup per bucket
 * gathers its runs and groups them in LDS, the first-seen numbering is a prefix count over one bit per entry, the neighbour
 * lookup probes a read-only image of the buckets' tables and lattice_offset is gathered per point - no global atomics per entry,
 * no scattered global stores.  Results are identical to efgh_lattice_level_build / _neighbors (same arrays, same meaning) except:
 *      list has efgh_lattice_part_list_len(n_cap, nbuckets) elements: a window of efgh_lattice_part_max_entries(n_cap) per
 *      bucket + an overflow area (vseg starts point into the windows);
 *      off, vseg, pts_next, vsid and info[SEG..] are written by the NEIGHBOURS call, not by the build.
 * nbuckets: power of two in [2, 8192], normally efgh_lattice_part_buckets(n_cap) (0: more points than the bucket limit - use the
 * hash build); slots: table slots per bucket, power of two in [16, 2048], >= ~2.5x the expected vertices per bucket.
 * big_buckets != 0: buckets with more than max_entries entries (lattice cells near the sensor of a real sweep collect thousands
 * of points) are grouped by a second kernel with 152 KB of LDS per bucket (up to 8 192 entries; one more launch).  A bucket
 * beyond what the call can hold, or more vertices than slots, sets bit 2 of info[EFGH_LATTICE_INFO_ERR] (nothing is written out
 * of bounds): rebuild with big_buckets, with more slots, or with the hash build.                                            */
int32_t efgh_lattice_part_max_entries(int32_t n_cap);
int64_t efgh_lattice_part_list_len(int32_t n_cap, int32_t nbuckets);
int32_t efgh_lattice_part_buckets(int32_t n_cap);
int64_t efgh_lattice_part_workspace_bytes(int32_t n_cap, int32_t h_cap, int32_t nsamples, int32_t nbuckets, int32_t slots);
/* bytes of `zeroed` (first-seen bitmap + an election counter): must be ALL ZERO when efgh_lattice_part_build is enqueued, and so
 * must info - so that one fill can serve every level of a pyramid */
int64_t efgh_lattice_part_zeroed_bytes(int32_t n_cap);
/* six launches (seven with big_buckets): keys + barycentric weights, key extrema, tile-local bucket sort, per-bucket grouping,
 * first-seen numbering */
int efgh_lattice_part_build(const float *pts, int64_t pts_cstride, const int32_t *n_dev, int32_t n_cap,
                            const int32_t *sid, int32_t pts_per_sample, int32_t nsamples, float scale32,
                            float *bary, float *emg, int32_t *list, int32_t h_cap, int32_t *info, void *workspace,
                            void *zeroed, int32_t nbuckets, int32_t slots, int32_t want_off, int32_t big_buckets, void *stream);
/* one launch: nbr + alist (as efgh_lattice_level_neighbors), off, and the vertex records vseg / pts_next / vsid / info[SEG..]
 * (pts ... div32 as passed to the build; pts_next has h_cap_build columns).  want_off = 0 in the build and off = NULL here skip
 * lattice_offset and the three arrays that only serve it (inference: the splat walks the vertex lists; off is needed by the
 * splat's backward). */
int efgh_lattice_part_neighbors(const void *workspace, const float *pts, int64_t pts_cstride, const int32_t *n_dev,
                                int32_t n_cap, const int32_t *sid, int32_t pts_per_sample, int32_t nsamples,
                                float scale32, float div32, int32_t h_cap_build, int32_t *info, int32_t h_cap,
                                int32_t *nbr, int32_t *alist, int32_t alias_cap, int32_t *off, int32_t *vseg,
                                float *pts_next, int32_t *vsid, int32_t nbuckets, int32_t slots, void *stream);

/* ------------------------------------------------------------------ BCL splat (K3) ---------
 * replaces SparseSum + density normalisation, nets/bilateralNN.py:6-40, 179-211, as a gather over the vertex lists of the
 * lattice build (no floating-point atomics, fixed summation order).  The input row of point p is
 * [emg[p] (4 channels; emg may be NULL: no such channels) | feat[p*ldf .. +Cf)]  ->
 * splat [H][C], C = (emg ? 4 : 0) + Cf  (row h = reference row h+1; the reference's all-zero row 0 is represented by
 * neighbour index -1), already multiplied by 1/(sum_bary + 1e-5) when `normalize` != 0 (args['bcn_use_norm'], bilateralNN.py:196-211;
 * 0: the plain sparse sum).  wsum [H] holds the density (kept for backward).
 * lanes_per_vertex: 64, 32 (two vertices per wave; needs C/4 <= 32) or 0 = chosen from avg_len (entries per vertex,
 * 4*n_in / H).  The mappings differ in the fp32 summation order only; a given mapping is bit-reproducible.                */
int efgh_splat_gather(const float *emg, const float *feat, int64_t ldf, int32_t Cf, const float *bary,
                      const int32_t *list, const int32_t *vseg, int32_t H, int32_t avg_len, int32_t lanes_per_vertex,
                      int32_t normalize, float *splat, float *wsum, void *stream);
/* backward w.r.t. feat: gfeat[p][c] = sum_r bary[p][r] / (wsum[off[p][r]] + 1e-5) * gsplat[off[p][r]][coff + c], c < Cf
 * (normalize == 0: without the density factor); gsplat [H][C] */
int efgh_splat_bwd(const float *gsplat, int32_t C, int32_t coff, const float *wsum, int32_t Cf, const float *bary,
                   const int32_t *off, int32_t n, float *gfeat, int64_t ldg, int32_t normalize, void *stream);
/* adjoint of the blur's neighbour gather (autograd of bilateralNN.py:240-242) through the lattice's own table:
 * dst[h][c] = sum over (m, t < 15) with nbr[m][t] == h of src[m][t*C + c]; src [H][15*C], dst [H][C] (overwritten).
 * n_alias = info + EFGH_LATTICE_INFO_ALIAS (device).                                                                */
int efgh_table_gather_transposed(const float *src, const int32_t *nbr, int32_t H, int32_t C, const int32_t *alist,
                                 const int32_t *n_alias, int32_t alias_cap, float *dst, void *stream);
/* the aliased part of the blur's data gradient (see efgh_gemm_desc.table_alias_mask): for every record (m*16 + t, target) of
 * alist  dx[target][c] += sum_n dy[m][n] * w[(n*C + c)*15 + t], c < C  - w = the blur weight in the reference layout
 * [N][C][15] (bilateralNN.py:107).  Records are applied in a fixed order (sorted, one owner block per target): no atomics. */
int efgh_blur_dgrad_alias(const float *dy, int64_t ldy, int32_t N, const float *w, int32_t C, const int32_t *alist,
                          const int32_t *n_alias, int32_t alias_cap, float *dx, int64_t ldx, void *stream);

/* ------------------------------------------------------------------ gather-GEMM (K4,K6,K8) -
 * One implicit-GEMM kernel family on fp32 MFMA (v_mfma_f32_32x32x2_f32):
 *     out[m][n] = act( scale[n]*(sum_{t<T,c<C} A[row(m,t)][c] * W[n][t*C+c] + bias[n])
 *                      + shift[n] + residual[m][n] )
 * replaces  Conv2d / ConvTranspose2d / Conv1d / Linear as used by nets/vgg.py:69-83,
 *           nets/resnet.py:55-71, nets/net_utils.py:35-98, nets/gnet.py, nets/fnet.py and the
 *           neighbour gather + blur_conv of nets/bilateralNN.py:240-246.                     */
typedef struct {
    /* A operand */
    const float *A;        /* rows of lda floats, first C used (channel slice = pointer offset) */
    int64_t lda;
    int32_t C;             /* channels per tap, multiple of 4 */
    int32_t T;             /* taps (<= 16) */
    int32_t mode;          /* 0 = direct rows (row = m), 1 = conv geometry, 2 = neighbour table,
                              3 = row stack: m = b*Wv + j, tap t = image row t (T = Hin, may exceed 16), the C
                              floats of a tap start at pixel j (sliding window; F correlation) */
    /* mode 1: m = (b*Hv + i)*Wv + j ;  input pixel (i*sh + dh[t], j*sw + dw[t]) of image b */
    int32_t B, Hin, Win, Hv, Wv, sh, sw;
    int8_t dh[16], dw[16];
    /* output pixel (i*osh + oh0, j*osw + ow0) of a [B][Ho][Wo] image (transposed conv classes) */
    int32_t Ho, Wo, osh, osw, oh0, ow0;
    /* mode 2: row = table[m*16 + t] (or -1 -> zeros) */
    const int32_t *table;
    /* W operand: packed [N][K], K = T*C */
    const float *W;
    int32_t N;
    int64_t M;
    const int32_t *M_dev;  /* optional: M read on device (grid sized by the M above) */
    /* epilogue */
    const float *bias, *scale, *shift;   /* each optional, [N] */
    const float *residual; int64_t ldr;  /* optional [M][ldr] */
    int32_t act;           /* 0 none, 1 relu, 2 leaky */
    float slope;
    float *out; int64_t ldo;
    float *stats;          /* optional [gridM][2][N] per-block column sum / sum of squares of the
                              pre-activation value (train-mode BatchNorm statistics) */
    /* optional batching: nbatch > 1 runs nbatch independent problems of identical shape in ONE launch;
     * problem z uses A + z*batch_stride_a, W + z*batch_stride_w, out + z*batch_stride_out (elements) */
    int32_t nbatch;
    int64_t batch_stride_a, batch_stride_w, batch_stride_out;
    /* mode 2 only: problem z reads table columns z*batch_stride_table .. + T (a split of the neighbour taps over the problems:
     * split-K for levels with few vertices, the partial planes are added by efgh_fold_planes) */
    int32_t batch_stride_table;
    /* mode 2 only: != 0 = taps whose bit is set in column 15 of the table row (the aliased neighbour hits marked by the lattice
     * build) read zeros.  What is left of the table is a symmetric relation, so the data gradient of the blur's neighbour gather
     * + convolution is this same gather-GEMM on the gradient with tap-mirrored weights (no [H][15 C] intermediate); the aliased
     * hits are added by efgh_blur_dgrad_alias. */
    int32_t table_alias_mask;
    /* stats_mode 1 (honoured by efgh_wino_conv3x3 and - round 6, bn_y == NULL only - efgh_wino2d_output): `stats` [rows][2][N] receives, instead of the forward statistics, the two
     * column sums the BatchNorm backward of the PRODUCER of A's gradient needs of the value this launch writes (a data gradient,
     * after the residual add):   sum g   and   sum g * (bn_raw - bn_mean) * bn_invstd,
     *     g = out * act'(bn_y ? bn_y : bn_raw * bn_pscale + bn_pshift)
     * where bn_raw / bn_y are the producer layer's raw convolution output / activation at the rows and columns of `out`.
     * efgh_bwd_finalize_f32 folds the rows (in float64); the layer's own reduction pass over dy and raw is then not needed. */
    int32_t stats_mode;
    const float *bn_raw; int64_t bn_ldraw;
    const float *bn_y; int64_t bn_ldy;
    const float *bn_pscale, *bn_pshift, *bn_mean, *bn_invstd;
    int32_t bn_act; float bn_slope;
} efgh_gemm_desc;

int efgh_gather_gemm(const efgh_gemm_desc *d, void *stream);
/* fold of the stats_mode-1 rows: sum_dpre [C], sum_dpre_xhat [C] (float) and their means over `count` rows (double) - what
 * efgh_act_bn_bwd_reduce returns */
int efgh_bwd_finalize_f32(const float *stats, int32_t rows, int32_t C, double count, float *sum_dpre, float *sum_dpre_xhat,
                          double *mean_dpre, double *mean_dpre_xhat, void *stream);
int32_t efgh_gather_gemm_grid_m(int64_t M, int32_t N);      /* rows of `stats` */

/* Wp[n][t][c] = W[n*sn + c*sc + tapidx[t]*st]   (weight re-layout for the kernel above) */
int efgh_pack_weight(const float *W, float *Wp, int32_t N, int32_t T, int32_t C, int64_t sn,
                     int64_t sc, int64_t st, const int32_t *tapidx_host, void *stream);
/* same, Wp[Np][T][Cp] with zeros outside [N][.][C] (row / channel counts rounded up for the kernels' vector width) */
int efgh_pack_weight_padded(const float *W, float *Wp, int32_t N, int32_t T, int32_t C, int32_t Np, int32_t Cp, int64_t sn,
                            int64_t sc, int64_t st, const int32_t *tapidx_host, void *stream);
/* the same for MANY weights in one launch (all packed layouts of a model after an optimizer step): jobs_dev = device array of
 * njobs records, prefix_dev [njobs + 1] = exclusive prefix sums of Np*T*Cp (int64), total = prefix_dev[njobs] */
typedef struct {
    const float *W; float *Wp;
    int32_t N, T, C, Np, Cp, pad_;
    int64_t sn, sc, st;
    int32_t taps[16];
} efgh_pack_job;
int efgh_pack_weight_batched(const efgh_pack_job *jobs_dev, const int64_t *prefix_dev, int32_t njobs, int64_t total, void *stream);
/* the same values through LDS tiles of 8 rows x 32 channels x T taps (coalesced on both sides): tile_prefix_dev [njobs + 1] =
 * exclusive prefix sums of ceil(Np / 8) * ceil(Cp / 32) (int64), ntiles = tile_prefix_dev[njobs] */
int efgh_pack_weight_batched_tiled(const efgh_pack_job *jobs_dev, const int64_t *tile_prefix_dev, int32_t njobs, int64_t ntiles,
                                   void *stream);
/* out[m][n] = act(sum_z part[z][m][n] + bias[n]), part [S][M][N] contiguous (N % 4 == 0): the planes of a split-K launch */
int efgh_fold_planes(const float *part, int32_t S, int64_t M, int32_t N, const float *bias, int32_t act, float slope, float *out,
                     int64_t ldo, void *stream);
/* out[i] = i < n ? v[i] : fill, i < np   (bias / BatchNorm vectors of layers whose width is not a multiple of 4) */
int efgh_pad_vec(const float *v, int32_t n, float *out, int32_t np, float fill, void *stream);

/* ------------------------------------------------------------------ BatchNorm / elementwise --
 * replaces nn.BatchNorm1d/2d + ReLU/LeakyReLU + residual add + nn.MaxPool2d(2,2) as composed in
 * nets/vgg.py:69-83, nets/resnet.py:55-71, nets/net_utils.py:45-98, nets/enet.py:150-152.     */

/* reduce the GEMM epilogue's per-block (sum,sumsq) partials [G][2][C] to the affine
 * y = x*scale + shift of train-mode BatchNorm; updates running_mean/var (momentum form, unbiased
 * variance) when rmean != NULL; save_mean/save_invstd (optional) are kept for backward.       */
int efgh_bn_finalize(const float *stats, int32_t G, int32_t C, double count, const float *gamma,
                     const float *beta, float *rmean, float *rvar, float momentum, float eps,
                     float *scale, float *shift, float *save_mean, float *save_invstd, void *stream);
/* same partials for a matrix that was not produced by the GEMM: stats [groups(M)][2][C] */
int32_t efgh_col_stats_groups(int64_t M);
int efgh_col_stats(const float *x, int64_t M, int32_t C, int64_t ld, float *stats, void *stream);
/* y[r][c] = act(x[r][c]*scale[c] + shift[c] + res[r][c]); scale/shift/res optional */
int efgh_scale_shift_act(const float *x, int64_t ldx, const float *scale, const float *shift,
                         const float *res, int64_t ldr, float *y, int64_t ldy, int64_t M, int32_t C,
                         int32_t act, float slope, void *stream);
/* the same pass, also leaving the sign bits of y (C % 32 == 0): bit (r*C + c) % 32 of word (r*C + c) / 32 = (y[r][c] > 0), M*C/32
 * words - the activation mask of a residual layer for efgh_act_bn_bwd_reduce / _apply (pass it as `y` with ldy = 0)       */
int efgh_scale_shift_act_bits(const float *x, int64_t ldx, const float *scale, const float *shift,
                              const float *res, int64_t ldr, float *y, int64_t ldy, uint32_t *bits, int64_t M, int32_t C,
                              int32_t act, float slope, void *stream);
int efgh_maxpool2(const float *x, float *y, int32_t B, int32_t H, int32_t W, int32_t C, void *stream);
/* the vertical half alone, [B][H][W][C] -> [B][H/2][W][C]: behind a producer that took the horizontal half (efgh_wino_conv3x3_hpool) */
int efgh_maxpool_v2(const float *x, float *y, int32_t B, int32_t H, int32_t W, int32_t C, void *stream);
/* model-boundary layout changes: (B,Cs,H,W) <-> [B][H][W][Cd] (extra channels zero) */
int efgh_nchw_to_nhwc(const float *x, float *y, int32_t B, int32_t Cs, int64_t HW, int32_t Cd, void *stream);
int efgh_nhwc_to_nchw(const float *x, int64_t ld, float *y, int32_t B, int32_t Cs, int64_t HW, void *stream);
/* torch.max(x, 2) over ragged segments of rows (enet.py:154, hnet.py:53): y[s][c], argrow opt. */
int efgh_segment_colmax(const float *x, int64_t ld, int32_t C, const int32_t *seg, int32_t nseg,
                        float *y, int32_t *argrow, void *stream);
/* torch.mean(x, 2) over equal segments (gnet.py:165) */
int efgh_segment_colmean(const float *x, int64_t ld, int32_t C, int32_t rows_per_seg, int32_t nseg,
                         float *y, void *stream);
/* the same two reductions in two stages (row slices per segment, folded in order: same maxima / first rows, the mean from another
 * summation order of the same doubles) - the forms the network uses: a segment is spread over ~1024 / nseg workgroups instead of
 * one per 256 (32) columns.  workspace: efgh_segment_workspace(C, nseg) bytes, 8-byte aligned; rows_hint = total rows (sizes the slices) */
int64_t efgh_segment_workspace(int32_t C, int32_t nseg);
int efgh_segment_colmax_ws(const float *x, int64_t ld, int32_t C, const int32_t *seg, int32_t nseg, int64_t rows_hint, float *y,
                           int32_t *argrow, void *workspace, void *stream);
int efgh_segment_colmean_ws(const float *x, int64_t ld, int32_t C, int32_t rows_per_seg, int32_t nseg, float *y, void *workspace,
                            void *stream);
/* nn.Softmax(dim=1) over 2 channels, written planar (B,2,H,W) (gnet.py:124) */
int efgh_softmax2_to_nchw(const float *x, int64_t ld, float *y, int32_t B, int64_t HW, void *stream);
/* G's depth and mask heads carried as ONE map x[B*HW][4] (channel 0 depth, channels 1-2 mask logits): g_depth (B,1,HW) and
 * g_mask = softmax over the two logits (B,2,HW) of gnet.py:121-124 from one read. */
int efgh_heads_to_nchw(const float *x, float *depth, float *mask, int32_t B, int64_t HW, void *stream);

/* ------------------------------------------------------------------ rasterisers / rotate ----
 * range image: common/torch_utils.py:11-59 applied to e_l.[pc;1] (nets/fnet.py:43-45).
 * pc [B][3][N], e_l [B][4][4] -> img [B][H][W][4] = (x,y,z,r); duplicates: last point wins.
 * pix [B][N] (pixel of each point or -1), vals [B][N][4] and winner [B][H*W] are caller scratch;
 * pix is what efgh_raster_bwd needs.                                                          */
int efgh_range_image(const float *pc, const float *e_l, int32_t B, int32_t N, int32_t H, int32_t W,
                     double fov_up, double fov_down, int32_t *pix, float *vals, int32_t *winner,
                     float *img, void *stream);
/* depth image: common/torch_utils.py:61-103; cam_T_velo [B][3][4]; img = (px,py,pz,w) */
int efgh_depth_image(const float *pc, const float *cam_T_velo, int32_t B, int32_t N, int32_t H,
                     int32_t W, int32_t *pix, float *vals, int32_t *winner, float *img, void *stream);
/* d(values)/d(img): gvals[b][i] = gimg[b][pix[b][i]] for every rasterised point */
int efgh_raster_bwd(const int32_t *pix, const float *gimg, int32_t B, int32_t N, int64_t HW,
                    float *gvals, void *stream);
/* gradient of the rasterised VALUES w.r.t. the pose that produced them, straight from the gradient image (no [B][N][4] pass):
 * mode 0 = range image, g_pose [B][16] = d/d e_l (needs e_l); mode 1 = depth image, g_pose [B][12] = d/d cam_T_velo (row 2 only).
 * partials: B * 64 * 16 doubles of scratch.  Replaces autograd through common/torch_utils.py:11-103 for the pose argument.  */
int efgh_raster_pose_bwd(const int32_t *pix, const float *gimg, const float *pc, const float *e_l, int32_t B, int32_t N,
                         int64_t HW, int32_t mode, double *partials, float *g_pose, void *stream);
/* PIL.Image.rotate(angle) NEAREST on uint8 (common/torch_utils.py:235-254): img (B,3,H,W) float
 * holding 0..255, rot_deg [B] degrees (fp32, as torch_utils.py:245 computes it);
 * out_nchw (B,3,H,W) and/or out_nhwc4 [B][H][W][4] (4th channel 0).                           */
int efgh_rotate_nearest_u8(const float *img, const float *rot_deg, int32_t B, int32_t H, int32_t W,
                           float *out_nchw, float *out_nhwc4, void *stream);

/* ------------------------------------------------------------------ F correlation head ------
 * nets/fnet.py:57,64,78-81 and circular_assign_torch, common/torch_utils.py:271-284.         */
int32_t efgh_minmax_groups(int64_t n);
/* per-sample global (min,max): x [B][n] -> mm [B][2]; part [B][groups][2] scratch */
int efgh_minmax(const float *x, int32_t B, int64_t n, float *part, float *mm, void *stream);
/* rp [B][h][wpitch][C] = pad(rng / (max-min)) : mirror on the left, circular on the right; row pitch
 * wpitch >= w+2*off, pixels past the padded width are zero */
int efgh_corr_pad(const float *rng, const float *rng_mm, int32_t B, int32_t h, int32_t w, int32_t C,
                  int32_t off, int32_t wpitch, float *rp, void *stream);
/* score [B][wp-wc+1] = sigmoid(corr / 16); cam [B][h][wc][16] is divided by (max-min) on the fly;
 * part [B][h][wp-wc+1] scratch; logit optional (pre-sigmoid, for tests / backward)           */
int efgh_corr1d(const float *rp, const float *cam, const float *cam_mm, int32_t B, int32_t h,
                int32_t wc, int32_t wp, float *part, float *logit, float *score, void *stream);

/* ------------------------------------------------------------------ backward (training) -----
 * The reference relies on torch autograd (iterater.py:42 `losses['total'].backward()`); these are
 * the hand-written adjoints of the kernels above.                                             */

/* weight gradient in the packed layout: dWp[n][t*C+c] = sum_m G[orow(m)][n] * A[row(m,t)][c];
 * `d` describes the SAME gather as the forward launch (A, lda, C, T, mode, geometry/table, N, M);
 * G is the gradient w.r.t. the (pre-BatchNorm) GEMM output, rows addressed like `out`.
 * The rows m are cut into chunks (one workgroup row per chunk); every chunk writes its own partial [N][K] plane into
 * `workspace` with plain stores and the planes are then added in chunk order: no atomics, no memset, the result is
 * bit-reproducible run to run.  efgh_gather_wgrad_workspace(d) = floats of `workspace` needed (0: a single chunk writes dWp
 * directly and `workspace` may be NULL); for the batched form set d->nbatch before asking.                               */
/* `out` (may be NULL): where the gradient belongs in the CALLER's own layout - W[n*sn + c*sc + taps[t]*st] (+= when accumulate) for
 * rows n < N, channels c < C of the packed [Np][T][Cp] gradient (Np = N rounded up to 4).  When the call's last launch can write
 * that layout itself (the fold of its row-chunk partials, or the finish kernel of a Winograd weight gradient) it does and the call
 * returns EFGH_WROTE_OUT with dWp untouched; otherwise (a single row chunk, taps the finish kernel cannot express) it returns EFGH_OK
 * with the packed gradient in dWp and the caller runs efgh_unpack_weight.  Nothing is remembered between calls. */
typedef struct efgh_wgrad_out_desc {
    float *W;
    int32_t N, T, C, Cp;
    int64_t sn, sc, st;
    int32_t taps[16];
    int32_t accumulate;
} efgh_wgrad_out_desc;
int64_t efgh_gather_wgrad_workspace(const efgh_gemm_desc *d);
int efgh_gather_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                      const efgh_wgrad_out_desc *out, void *stream);
/* the same contraction for d->nbatch independent problems in ONE launch (mode 0 only): problem b reads A + b*d->batch_stride_a
 * and G + b*batch_stride_g and writes dWp + b*batch_stride_dw (= N*C) */
int efgh_gather_wgrad_batched(const efgh_gemm_desc *d, const float *G, int64_t ldg, int64_t batch_stride_g, float *dWp,
                              int64_t batch_stride_dw, float *workspace, void *stream);

/* ------------------------------------------------------------------ Winograd F(4x4,3x3) --------
 * the 3x3 / stride 1 / pad 1 convolutions with >= 128 channels (nets/vgg.py:77, nets/resnet.py:22-30), their data and weight
 * gradients, as input transform -> 36 batched GEMMs (efgh_gather_gemm, mode 0) -> output transform; fp32 throughout.
 * `d` is the mode-1 descriptor of the layer (efgh_wino2d_supported: N/2 and C/2 powers of two <= 256).
 *   T = efgh_wino2d_tiles(B,H,W) 4x4 output tiles;  V [T][36][C],  U [36][N][C],  M [T][36][N],  Gy [T][36][N],  S [36][N][C]
 *   (tile-major activations: GEMM a uses rows of length C at stride 36*C, batch stride C).
 *   efgh_wino2d_output applies d's epilogue (bias, stats [efgh_wino2d_stats_rows][2][N], scale/shift, residual, act) -> d->out. */
int efgh_wino2d_supported(const efgh_gemm_desc *d);
int64_t efgh_wino2d_tiles(int32_t B, int32_t H, int32_t W);
int32_t efgh_wino2d_stats_rows(int32_t B, int32_t H, int32_t W, int32_t N);
int efgh_wino2d_pack(const float *Wp, float *U, int32_t N, int32_t C, void *stream);          /* U = G w G^T of packed [N][9][C] */
/* every Winograd-domain weight of a model in one launch: kind 0 = efgh_wino_pack (1-D F(4,3), C % 16 == 0), kind 1 =
 * efgh_wino2d_pack; jobs_dev: DEVICE array sorted by first_block, job j owns the workgroups [first_block[j], first_block[j+1]) of
 * 256 work items each (3*C*N items for kind 0, N*C for kind 1), nblocks = their total */
typedef struct efgh_wino_pack_job {
    const float *Wp;
    float *U;
    int32_t N, C, kind, pad_;
    int64_t first_block;
} efgh_wino_pack_job;
int efgh_wino_pack_batched(const efgh_wino_pack_job *jobs_dev, int32_t njobs, int64_t nblocks, void *stream);
int efgh_wino2d_input(const float *A, int64_t lda, int32_t C, int32_t B, int32_t H, int32_t W, float *V, void *stream);
/* round 6: the same transform of act(A*scale[c] + shift[c]) - A is the RAW output of a train-mode BatchNorm layer whose normalise +
 * activate pass (efgh_scale_shift_act) was not run because this layer is the activation's only consumer (nets/resnet.py:55-71
 * conv1 -> conv2; nets/vgg.py:69-83 un-pooled pairs; nets/net_utils.py:66-98 convT -> conv): one write + one read of it less */
int efgh_wino2d_input_act(const float *A, int64_t lda, int32_t C, int32_t B, int32_t H, int32_t W, const float *scale,
                          const float *shift, int32_t act, float slope, float *V, void *stream);
/* round 6: the "apply" pass of the BatchNorm backward (efgh_act_bn_bwd_apply) inside the two gradient-side transforms of a 2-D Winograd
 * layer: from dy, raw, the activation mask (ybits: the sign bits efgh_scale_shift_act_bits left, N % 32 == 0; or pscale / pshift: the
 * mask is re-derived from raw*pscale + pshift - exactly one of the two), mean / invstd / coef [N] and the finalised float64 means m1 / m2
 * of efgh_act_bn_bwd_reduce it computes draw = coef*(dy*act' - m1 - xhat*m2) on the fly and leaves Vd = B^T draw B [T][36][N] (what
 * efgh_wino2d_input(draw) writes: the data gradient's operand), Gy = G4 draw G4^T [T][36][N] (efgh_wino2d_dy(draw): the weight
 * gradient's) and - when dres != NULL - dres = dy*act' (the gradient of the residual branch).  draw itself is never stored. */
int efgh_wino2d_bwd_transforms(const float *dy, int64_t lddy, const float *raw, int64_t ldraw, const uint32_t *ybits,
                               const float *pscale, const float *pshift, const float *mean, const float *invstd, const float *coef,
                               const double *m1, const double *m2, int32_t N, int32_t B, int32_t H, int32_t W, int32_t act,
                               float slope, float *Vd, float *Gy, float *dres, int64_t lddres, void *stream);
/* the same for a layer whose activation went straight into MaxPool2d(2,2) (nets/vgg.py:69-83; efgh_maxpool2_affine in the forward pass):
 * dy_pool [B][H/2][W/2][N] is the POOLED gradient; dpre = dy_pool * act' at the first maximum (scan order) of every 2x2 window of
 * act(raw*pscale + pshift), 0 elsewhere (what efgh_pool_bn_bwd_apply evaluates); m1 / m2 from efgh_pool_bn_bwd_reduce */
int efgh_wino2d_bwd_transforms_pooled(const float *dy_pool, int64_t lddy, const float *raw, int64_t ldraw, const float *pscale,
                                      const float *pshift, const float *mean, const float *invstd, const float *coef,
                                      const double *m1, const double *m2, int32_t N, int32_t B, int32_t H, int32_t W, int32_t act,
                                      float slope, float *Vd, float *Gy, void *stream);
int efgh_wino2d_output(const float *M, const efgh_gemm_desc *d, void *stream);
/* the same output transform for a layer that is followed by nn.MaxPool2d(2,2) (nets/vgg.py:69-83, inference): d->out is the POOLED map
 * [B][Hin/2][Win/2][ldo] = max over each 2x2 window of act((v + bias)*scale + shift); no residual, no statistics */
int efgh_wino2d_output_pooled(const float *M, const efgh_gemm_desc *d, void *stream);
int efgh_wino2d_dy(const float *G, int64_t ldg, int32_t N, int32_t B, int32_t H, int32_t W, float *Gy, void *stream);
int efgh_wino2d_wfinish(const float *S, float *dWp, int32_t N, int32_t C, const efgh_wgrad_out_desc *out, void *stream);   /* dWp [N][9][C] = A3^T S A3 */

/* W.flat[n*sn + c*sc + tapidx[t]*st] (+)= Wp[n][t][c]  (Wp rows padded to Cp) */
int efgh_unpack_weight(const float *Wp, float *W, int32_t N, int32_t T, int32_t C, int32_t Cp, int64_t sn,
                       int64_t sc, int64_t st, const int32_t *tapidx_host, int32_t accumulate, void *stream);
/* dst[table[m*16+t]][c] += src[m][t*C+c]  (adjoint of the neighbour gather, bilateralNN.py:240-242) */
int efgh_table_scatter_add(const float *src, const int32_t *table, int64_t M, int32_t T, int32_t C,
                           float *dst, void *stream);
/* BatchNorm(+residual)+activation backward, two passes.
 * reduce: dpre = dy*act'(y); sum_dpre[c], sum_dpre_xhat[c] (= dbeta, dgamma) and their means;
 *         mean/invstd/raw NULL -> only sum_dpre (bias gradient).  part: [efgh_bwd_groups(M)][2][C] float64 (the column sums and the two means are kept in double, as torch's CPU kernel does).
 *         y == NULL: the activation mask is recomputed from raw*pscale + pshift (layers without a
 *         residual), which saves one full read of the activation in both passes.
 *         ldy == 0: `y` points to the SIGN BITS of the activation (uint32 words, efgh_scale_shift_act_bits; C % 32 == 0) - residual
 *         layers, whose mask cannot be re-derived from raw alone: 1/32 of the bytes of reading the activation back.            */
int32_t efgh_bwd_groups(int64_t M);
int efgh_act_bn_bwd_reduce(const float *dy, int64_t lddy, const float *y, int64_t ldy, const float *raw,
                           int64_t ldraw, const float *mean, const float *invstd, const float *pscale,
                           const float *pshift, int64_t M, int32_t C,
                           int32_t act, float slope, double *part, float *sum_dpre, float *sum_dpre_xhat,
                           double *mean_dpre, double *mean_dpre_xhat, void *stream);
/* apply: draw = coef*(dpre - m1 - xhat*m2)  (train BN)  |  coef*dpre (eval BN / none); dres = dpre */
int efgh_act_bn_bwd_apply(const float *dy, int64_t lddy, const float *y, int64_t ldy, const float *raw,
                          int64_t ldraw, const float *mean, const float *invstd, const float *coef,
                          const double *m1, const double *m2, const float *pscale, const float *pshift,
                          int64_t M, int32_t C, int32_t act, float slope,
                          float *draw, int64_t lddraw, float *dres, int64_t lddres, void *stream);
int efgh_maxpool2_bwd(const float *x, const float *dy, float *dx, int32_t B, int32_t H, int32_t W, int32_t C,
                      void *stream);
/* BatchNorm + activation + MaxPool2d(2,2) fused over the raw conv output (training path of the VGG trunks, nets/vgg.py:69-83):
 * y = max over the window of act(x*scale + shift); the backward recomputes the activated window and routes dy to its first
 * maximum (rows / columns beyond 2*floor(H/2), 2*floor(W/2) are not written: zero dx beforehand when H or W is odd).        */
int efgh_maxpool2_affine(const float *x, const float *scale, const float *shift, int32_t act, float slope, float *y,
                         int32_t B, int32_t H, int32_t W, int32_t C, void *stream);
int efgh_maxpool2_bwd_affine(const float *x, const float *scale, const float *shift, int32_t act, float slope,
                             const float *dy, float *dx, int32_t B, int32_t H, int32_t W, int32_t C, void *stream);
/* BatchNorm backward of such a fused layer straight from the POOLED gradient dy_pool [B][H/2][W/2][C] (train-mode BN): the 2x2
 * window is recomputed from raw, the pooled gradient goes to its first maximum, all other elements have dpre = 0.  Same outputs
 * as efgh_act_bn_bwd_reduce / _apply (means over all B*H*W positions); part: [efgh_pool_bwd_groups(B,H,W)][2][C] float64.      */
int32_t efgh_pool_bwd_groups(int32_t B, int32_t H, int32_t W);
int efgh_pool_bn_bwd_reduce(const float *dy_pool, const float *raw, const float *mean, const float *invstd, const float *pscale,
                            const float *pshift, int32_t B, int32_t H, int32_t W, int32_t C, int32_t act, float slope,
                            double *part, float *sum_dpre, float *sum_dpre_xhat, double *mean_dpre, double *mean_dpre_xhat,
                            void *stream);
/* round 6: the same sums for a ReLU layer from POOLED tensors only - the pooled gradient and the pooled activation y_pool
 * [B][H/2][W/2][C] (dpre is non-zero only where y_pool > 0, and there xhat = (y_pool - beta) / gamma of the winning element): half a
 * unit of traffic instead of 1.25.  raw is read only for channels with pscale == 0.  part: [efgh_bwd_groups(B*(H/2)*(W/2))][2][C] */
int efgh_pool_bn_bwd_reduce_pooled(const float *dy_pool, const float *y_pool, const float *raw, const float *mean,
                                   const float *invstd, const float *pscale, const float *pshift, int32_t B, int32_t H, int32_t W,
                                   int32_t C, double *part, float *sum_dpre, float *sum_dpre_xhat, double *mean_dpre,
                                   double *mean_dpre_xhat, void *stream);
int efgh_pool_bn_bwd_apply(const float *dy_pool, const float *raw, const float *mean, const float *invstd, const float *coef,
                           const double *m1, const double *m2, const float *pscale, const float *pshift, int32_t B, int32_t H,
                           int32_t W, int32_t C, int32_t act, float slope, float *draw, void *stream);
int efgh_segment_colmax_bwd(const float *dy, const int32_t *argrow, int32_t nseg, int32_t C, float *dx,
                            int64_t ld, void *stream);
int efgh_segment_colmean_bwd(const float *dy, int32_t P, int32_t nseg, int32_t C, float *dx, int64_t ld,
                             void *stream);
int efgh_softmax2_bwd(const float *y, const float *dy, int32_t B, int64_t HW, float *dx, int64_t ld, void *stream);
/* backward of efgh_heads_to_nchw: y = the softmax output; dmask (B,2,HW) / ddepth (B,1,HW) may be NULL; dx[B*HW][4]. */
int efgh_heads_bwd(const float *y, const float *dmask, const float *ddepth, int32_t B, int64_t HW, float *dx, void *stream);
/* correlation head: dcam_n (gradient w.r.t. cam/(max-min)) and drp (w.r.t. the padded, normalised
 * range features); efgh_corr_unpad folds drp back onto the un-padded map.                      */
int efgh_corr1d_bwd(const float *rp, const float *cam, const float *cam_mm, const float *dlogit, int32_t B,
                    int32_t h, int32_t wc, int32_t wp, float *dcam_n, float *drp, void *stream);
int efgh_corr_unpad(const float *drp, int32_t B, int32_t h, int32_t w, int32_t C, int32_t off, float *dx,
                    void *stream);

/* fused Adam over ONE flat fp32 buffer holding all 47.8 M parameters (replaces the 353 per-tensor
 * updates of torch.optim.Adam, main.py:181-183; lr schedule of common/helper.py:28-38 is the
 * caller's).  g is multiplied by grad_scale first (1/world for a summed all-reduce).          */
int efgh_adam_step(float *w, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void *stream);

/* "thin" layers (<= 4 channels on one side: RGB/range/depth input convs, the 1-/2-channel heads and
 * their dgrad/wgrad): HBM-bound VALU kernels with the descriptor, gather modes and epilogue of
 * efgh_gather_gemm (mode 1 only, no `stats`).  efgh_thin_supported: 0 = no, 1 = C==4 form, 2 = N==4 (VALU), 3 = N==4 on MFMA
 * (64 channels, 3x3, stride 1, pad 1: k_n4_conv3x3_c64), 4 = the 4 -> 4 channel 3x3 stencil (k_c4n4_conv3x3). */
int efgh_thin_supported(const efgh_gemm_desc *d);
int efgh_thin_gemm(const efgh_gemm_desc *d, void *stream);
/* weight gradient of the same thin layers (C == 4 with N/4 a power of two and T in {1, 2, 4, 9}; or N == 4): no atomics - every
 * workgroup leaves a partial [N][T*C] plane in `workspace` (efgh_thin_wgrad_workspace floats), folded in a fixed order into dWp. */
int64_t efgh_thin_wgrad_workspace(const efgh_gemm_desc *d);
int efgh_thin_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                    const efgh_wgrad_out_desc *out, void *stream);
/* 4 -> 4 channels (the 1- / 2-channel convolutions behind G's transposed heads, gnet.py:56-68), 3x3, stride 1, pad 1: weight gradient
 * without atomics - one partial [4][9][4] plane per workgroup in `workspace` (efgh_c4n4_wgrad_workspace floats), folded in a fixed
 * order into dWp. */
int efgh_c4n4_supported(const efgh_gemm_desc *d);
int64_t efgh_c4n4_wgrad_workspace(const efgh_gemm_desc *d);
int efgh_c4n4_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                    const efgh_wgrad_out_desc *out, void *stream);

/* 3x3 convolutions with C == 4 input channels per tap, stride 1 or 2, pad 1, N in {32, 64, 128} on fp32 MFMA (the RGB / range /
 * depth input layers: nets/vgg.py:77 first conv, nets/gnet.py:21,80; and the data gradient of G's transposed heads): the three
 * input rows of 128 output pixels are staged in LDS once, the 36 x N weights stay in registers of a persistent workgroup.  Same
 * descriptor and epilogue as efgh_gather_gemm; `stats` has efgh_c4_stats_rows(B, Ho, Wo) rows.  efgh_c4_wgrad: the weight
 * gradient dWp [N][9][4] of the same layers (N % 16 == 0, 32 <= N <= 128) on v_mfma_f32_16x16x4_f32, G [M][ldg] read once; one
 * partial plane per wave in `workspace` (efgh_c4_wgrad_workspace floats), folded in a fixed order: no atomics.                  */
/* 3x3 / stride-1 / pad-1 layers with 16 or 32 channels on both sides (csrc/smallc.hip: F's up-sampling stages, nets/fnet.py:22-31):
 * same descriptor and epilogue as efgh_gather_gemm, W packed [N][9][C]; `stats` has efgh_sc_stats_rows(B, H, W) rows.
 * efgh_sc_wgrad: dWp [N][9][C], workspace of efgh_sc_wgrad_workspace(d) floats (one partial plane per wave, folded in a fixed
 * order: no atomics). */
int efgh_sc_supported(const efgh_gemm_desc *d);
int32_t efgh_sc_stats_rows(int32_t B, int32_t H, int32_t W);
int efgh_sc_conv3x3(const efgh_gemm_desc *d, void *stream);
int efgh_sc_wgrad_supported(const efgh_gemm_desc *d);      /* also C == 4 with N in {32, 64} at stride 1 (the RGB / range / depth input layers) */
int64_t efgh_sc_wgrad_workspace(const efgh_gemm_desc *d);
int efgh_sc_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                  const efgh_wgrad_out_desc *out, void *stream);
int efgh_c4_supported(const efgh_gemm_desc *d);
int32_t efgh_c4_stats_rows(int32_t B, int32_t Ho, int32_t Wo);
int efgh_c4_conv3x3(const efgh_gemm_desc *d, void *stream);
/* the same layer followed by nn.MaxPool2d(2,2) (nets/vgg.py:69-83, inference): stride 1, no residual / statistics; d->out is the
 * POOLED map [B][Ho/2][Wo/2][ldo]; bit-identical to efgh_c4_conv3x3 + efgh_maxpool2 */
int efgh_c4_pooled_supported(const efgh_gemm_desc *d);
int efgh_c4_conv3x3_pooled(const efgh_gemm_desc *d, void *stream);
int efgh_c4_wgrad_supported(const efgh_gemm_desc *d);
int64_t efgh_c4_wgrad_workspace(const efgh_gemm_desc *d);
int efgh_c4_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *dWp, float *workspace,
                  const efgh_wgrad_out_desc *out, void *stream);

/* debugging aid: the resident-workgroups-per-CU figures the weight-gradient chunking uses, as text */
int efgh_debug_occupancy(char *out, int32_t cap);

/* efgh_gather_gemm serves modes 0 / 1 with more than 32 outputs and C % 32 == 0 on LDS-DMA staged instances of its kernel since
 * round 5 (`global_load_lds_dwordx4` into a two-slot XOR-swizzled ring, one barrier per 32-deep step; same products, same k
 * order, same epilogue: bit-identical outputs).  This switch exists for tests and A/B timing: 0 = register-staged kernel for
 * every launch, 1 = default.  Returns the previous setting; process-wide. */
int efgh_gather_gemm_set_dma(int32_t on);

/* ------------------------------------------------------------------ batched plain GEMMs, LDS-DMA staged (round 5) ------
 * The 36 alpha planes of a 2-D Winograd F(4x4,3x3) layer (every 3x3 / stride-1 convolution with >= 256 channels, weight
 * gradient >= 128: nets/vgg.py:77, nets/resnet.py:22-30) are plain GEMMs over contiguous fp32 rows.  These entry points run
 * them with `global_load_lds_dwordx4` staging into an NBUF-slot LDS ring (one workgroup barrier per 32-deep step, no VGPR /
 * ds_write staging pass) instead of efgh_gather_gemm / efgh_gather_wgrad_batched (mode 0); same descriptor, same arithmetic
 * order, bit-identical results.  nbuf: ring slots, 2 or 3 (0 = default).
 *   efgh_plane_gemm:  out[b][m][n] = sum_k A[b][m][k] * W[b][n][k];  needs mode 0, T 1, C % 32 == 0, N % 128 == 0, no epilogue.
 *   efgh_plane_wgrad_batched: dWp[b][n][c] = sum_m G[b][m][n] * A[b][m][c];  needs C % 128 == 0, N % 128 == 0; row chunks leave
 *   partial planes in `workspace` (efgh_plane_wgrad_workspace floats), folded in chunk order.                              */
int efgh_plane_gemm_supported(const efgh_gemm_desc *d);
int efgh_plane_gemm(const efgh_gemm_desc *d, int32_t nbuf, void *stream);
int efgh_plane_wgrad_supported(const efgh_gemm_desc *d, int64_t ldg);
int64_t efgh_plane_wgrad_workspace(const efgh_gemm_desc *d);
int efgh_plane_wgrad_batched(const efgh_gemm_desc *d, const float *G, int64_t ldg, int64_t batch_stride_g, float *dWp,
                             int64_t batch_stride_dw, float *workspace, int32_t nbuf, void *stream);

/* stride-2 3x3 transposed conv with <= 4 output channels (G's depth / mask heads, gnet.py:56-68) as
 * ONE gather-GEMM over the input pixels (Y [B*Hin*Win][ldy], column (kh*3+kw)*O+o) + this fold:
 * out[b][oh][ow][o] = act(scale[o]*sum_{oh=2ih-pad+kh, ow=2iw-pad+kw} Y[b][ih][iw][..] + shift[o]).
 * efgh_convt_im2col is the adjoint unfold of the output gradient (feeds the weight-gradient GEMM). */
int efgh_convt_col2im(const float *Y, int64_t ldy, int32_t B, int32_t Hin, int32_t Win, int32_t Ho, int32_t Wo,
                      int32_t O, int32_t pad, const float *scale, const float *shift, int32_t act, float slope,
                      float *out, int64_t ldo, void *stream);
int efgh_convt_im2col(const float *G, int64_t ldg, int32_t B, int32_t Hin, int32_t Win, int32_t Ho, int32_t Wo,
                      int32_t O, int32_t pad, float *Ycol, int64_t ldy, void *stream);

/* MFMA formulation of the F correlation: the camera row is cut into nseg segments of segw pixels;
 * the image rows into nsplit groups of T = h/nsplit (split-K).  efgh_corr_pack_cam builds
 * Wc [B][nsplit][nseg][T][segw*16] (normalised, zero past the camera width); ONE batched
 * efgh_gather_gemm (mode 3, nbatch = B*nsplit) gives P[b][ks][m][s] = sum_{y in group,x<segw,c}
 * rp[b][y][m+x][c]*Wc[..], and efgh_corr_fold sums score[b][j] = sigmoid((1/16) sum_{ks,s} P[b][ks][j+s*segw][s]). */
int efgh_corr_pack_cam(const float *cam, const float *cam_mm, int32_t B, int32_t h, int32_t wc, int32_t segw,
                       int32_t nseg, int32_t nsplit, float *Wc, void *stream);
int efgh_corr_fold(const float *P, int32_t B, int32_t nsplit, int64_t Mv, int32_t ldp, int32_t nseg, int32_t segw,
                   int32_t nj, float *logit, float *score, void *stream);
/* MFMA formulation of the correlation's backward (adjoint of fnet.py:78-81 w.r.t. both feature maps): with the operands as
 * planes [(y,c)][position] (efgh_corr_planes; `mm` != NULL also applies the 1/(max-min) normalisation) and the Toeplitz matrix
 * of d(logit) (efgh_corr_toeplitz: T[r][c] = dl[c-r], or dl[r-c] when transpose), both gradients are batched mode-0
 * efgh_gather_gemm products; efgh_corr_unplanes restores the [y][position][c] layout.                                          */
/* backward of the per-sample normalisation x/(max-min) (fnet.py:57,64): dx = dxn/d -/+ S/(d^2 k) at the elements attaining
 * max / min (S = sum dxn*x, k = number of ties, as torch's max()/min() backward); part has 3*efgh_minmax_groups(n) floats per sample */
int efgh_norm_bwd(const float *x, const float *dxn, const float *mm, int32_t B, int64_t n, float *part, float *dx, void *stream);
int efgh_corr_planes(const float *x, const float *mm, int32_t B, int32_t h, int32_t w, int32_t w_in_pitch, int32_t wP,
                     float *out, void *stream);
int efgh_corr_unplanes(const float *in, int32_t B, int32_t h, int32_t w, int32_t wP, float *out, void *stream);
int efgh_corr_toeplitz(const float *dl, int32_t B, int32_t nj, int32_t rows, int32_t cols, int32_t colsP, int32_t transpose,
                       float *T, void *stream);

/* Winograd F(4,3) (along the image row) form of the "same" 3x3 / stride-1 convolutions of nets/vgg.py:77 and
 * nets/resnet.py:22-30 (and of their data gradients): six GEMMs of depth 3C over 4-pixel tiles, half the MFMA
 * work of efgh_gather_gemm on the same layer, all operands and accumulation in fp32.  `d` is the mode-1
 * descriptor of the layer (T = 9, taps (t/3-1, t%3-1), stride 1, C % 16 == 0, N % 64 == 0); d->W is ignored and
 * U = efgh_wino_pack(packed weight [N][9][C]) is used instead.  Same epilogue as efgh_gather_gemm; `stats` has
 * efgh_wino_grid_m(B, H, W) rows.  efgh_wino_supported: 1 if `d` qualifies.                               */
int efgh_wino_supported(const efgh_gemm_desc *d);
int32_t efgh_wino_grid_m(int32_t B, int32_t H, int32_t W);
int efgh_wino_pack(const float *Wp, float *U, int32_t N, int32_t C, void *stream);
/* inference, a layer followed by nn.MaxPool2d(2,2) (nets/vgg.py:69-83): d->out is the map of HALF the width [B][Hin][Win/2][ldo] =
 * max over horizontal pixel pairs of act((v + bias)*scale + shift); efgh_maxpool_v2 finishes the window.  No residual / statistics */
int efgh_wino_conv3x3_hpool(const efgh_gemm_desc *d, const float *U, void *stream);
int efgh_wino_conv3x3(const efgh_gemm_desc *d, const float *U, void *stream);
/* weight gradient of the same layers (replaces efgh_gather_wgrad for them; additionally C % 64 == 0): Winograd
 * F(3,4) over 4-pixel gradient tiles, six tile-contracted GEMMs per tile range written as partials [6][N][3C] into the
 * scratch S (efgh_wino_wgrad_workspace(d) floats; plain stores), then added in range order and folded into the packed
 * dWp [N][9][C]: bit-reproducible.                                                                          */
int efgh_wino_wgrad_supported(const efgh_gemm_desc *d);
int64_t efgh_wino_wgrad_workspace(const efgh_gemm_desc *d);
int efgh_wino_wgrad(const efgh_gemm_desc *d, const float *G, int64_t ldg, float *S, float *dWp, const efgh_wgrad_out_desc *out,
                    void *stream);

/* ------------------------------------------------------------------ sample preparation --
 * GPU form of the reference's per-sample CPU transforms, data_loader/loader_utils.py:104-202 (called from
 * kitti_odom_loader.py:251-273 / rellis3d_loader.py:306-339).  Images are (H,W,3) uint8, device resident.  The O(1)
 * geometry is host work (efgh_amd/data/prepare.py restates Pillow's Image.rotate / Resample.c coefficient code).   */
/* Image.rotate(angle, expand) NEAREST (common/numpy_utils.py:426-445): xin = (c[2]+c[0]*x+c[1]*y)>>16,
 * yin = (c[5]+c[3]*x+c[4]*y)>>16 (Pillow's 16.16 affine_fixed), zero fill; out is (nh,nw,3).                       */
int efgh_prep_affine_nearest_u8(const uint8_t *in, int32_t h, int32_t w, const int64_t *coef6_host, uint8_t *out,
                                int32_t nh, int32_t nw, void *stream);
/* crop_image / zero_pad_image (numpy_utils.py:447-503): out[y][x] = in[y+oy][x+ox] or 0                            */
int efgh_prep_crop_pad_u8(const uint8_t *in, int32_t h, int32_t w, int32_t oy, int32_t ox, uint8_t *out, int32_t th,
                          int32_t tw, void *stream);
/* one pass of Image.resize's 8-bit resampler (numpy_utils.py:474-486; Pillow Resample.c): along axis (1 = x, 0 = y),
 * out = clip8((2^21 + sum_k in[bounds[2o]+k]*coeffs[o][k]) >> 22), k < bounds[2o+1]; tables are device int32        */
int efgh_prep_resample_u8(const uint8_t *in, int32_t h, int32_t w, int32_t axis, int32_t out_size, const int32_t *bounds,
                          const int32_t *coeffs, int32_t ksize, uint8_t *out, void *stream);
/* (H,W,3) -> (3,H,W) uint8 and/or image_valid_mask (1,H,W) (numpy_utils.py:505-517)                                */
int efgh_prep_hwc_to_chw_u8(const uint8_t *in, int32_t h, int32_t w, uint8_t *chw, uint8_t *mask, void *stream);
/* network input: zero pad to (th,tw) at (oy,ox), uint8 -> float32, (3,th,tw) (loader_utils.py:110-114)              */
int efgh_prep_u8_to_f32_chw_pad(const uint8_t *in, int32_t h, int32_t w, int32_t oy, int32_t ox, float *out, int32_t th,
                                int32_t tw, void *stream);
/* preproc_pcd (loader_utils.py:160-202): stable compaction of the points with -r <= x,y < r (pcd [n][4] fp32,
 * optionally pre-gathered by pre_idx (lidar-line reduction) and with x,y negated (RELLIS axis flip));
 * keep_idx [n] receives the surviving source rows in order, *count their number;
 * block_scratch has efgh_prep_filter_blocks(n) ints.                                                               */
int32_t efgh_prep_filter_blocks(int32_t n);
int efgh_prep_radius_filter(const float *pcd, const int32_t *pre_idx, int32_t n, int32_t flip_xy, float radius,
                            int32_t *block_scratch, int32_t *keep_idx, int32_t *count, void *stream);
/* out[r][j] = rand_init_l[r] . (x,y,z,1) in float64 for point keep_idx[sel[j]] (sel NULL: j), j < n_sel; the remaining
 * columns are the transform of (0,0,0,1).  T34 = 12 doubles (device).  out32 (3,num_points) fp32 and/or out64.      */
int efgh_prep_gather_transform(const float *pcd, const int32_t *keep_idx, const int32_t *sel, int32_t n_sel,
                               int32_t flip_xy, const double *T34, int32_t num_points, float *out32, double *out64,
                               void *stream);

/* ------------------------------------------------------------------ evaluation metrics --
 * Err.calc_error_odom_np (mode 0: arccos((tr(pred_R^T gt_R)-1)/2) in degrees, |pred_t - gt_t|_2; common/helper.py:198-207)
 * and Err.calc_error_raw_np (mode 1: quaternion distance 2*atan2(|v|,|w|) of gt*pred^-1 in degrees, mean |dt|;
 * helper.py:165-196) for B pairs of row-major 4x4 poses on the device.                                           */
int efgh_pose_errors(const float *gt, const float *pred, int32_t B, int32_t mode, float *rot_err, float *trs_err,
                     void *stream);

/* ------------------------------------------------------------------ G image losses -----------------
 * losses/loss_utils.py:186-199: l_depth = sum(valid*(gt_depth - pred_depth)^2)/sum(valid), valid = gt_depth > 0 & img_mask > 0;
 * l_mask = mean BCE(pred_mask[:,0], gt_depth > 0) (torch's -100 log clamp).  gdep4 is efgh_depth_image's [B][H][W][4] output
 * (depth = channel 3); gt_depth / gt_mask (B,1,H,W) are written for the gt dict; part: 3*efgh_gimg_loss_groups(B*HW) doubles;
 * out3 = {l_depth, l_mask, sum(valid)} on the device.  The backward call writes both prediction gradients given the two
 * upstream scalars (device pointers).  *_bstride: batch stride in floats of channel 0 of the (B,2,H,W) mask tensors.        */
int32_t efgh_gimg_loss_groups(int64_t n);
int efgh_gimg_loss_fwd(const float *pred_depth, const float *pred_mask, int64_t mask_bstride, const float *gdep4,
                       const uint8_t *img_mask, int32_t B, int64_t HW, float *gt_depth, float *gt_mask, double *part,
                       float *out3, void *stream);
int efgh_gimg_loss_bwd(const float *pred_depth, const float *pred_mask, int64_t mask_bstride, const float *gt_depth,
                       const uint8_t *img_mask, int32_t B, int64_t HW, const float *sums3, const float *g_depth,
                       const float *g_mask, float *d_pred_depth, float *d_pred_mask, int64_t dmask_bstride, void *stream);

/* ---- TensorBoard / evaluation overlay images (common/numpy_utils.py:8-413) -------------------------------
 * efgh_sum_depth_last / efgh_sum_range_last: the float64 rasterisers of numpy_utils.py:338-358 / :299-336 for ONE sample
 *   (pc: three rows of N floats, pc_cstride apart; T34: the 3x4 matrix - for the range image the top three rows of the 4x4 -
 *   as float64): the LAST point of the sweep on a pixel wins.  depth: uint8 [H][W] (float64 -> uint8 wrap as astype does);
 *   range: float64 [H][W] of sqrt(x^2+y^2+z^2+1).  idx_ws: H*W int32 of scratch.
 * efgh_sum_paint: minmax_color_img_from_img_numpy's raster-order painting (:384-396) for njobs independent images; jobs_dev is
 *   a device array of {const double *in (normalised image), double *out (ZEROED by the caller), int32 H, W, px}.
 * efgh_sum_colorize: matplotlib's look-up (int(x*256), x == 1 -> 255) through a uint8 [256][3] table, plus the != 0 mask. */
int efgh_sum_depth_last(const float *pc, int64_t pc_cstride, int32_t N, const double *T34, int32_t H, int32_t W, int32_t *idx_ws,
                        uint8_t *out, void *stream);
int efgh_sum_range_last(const float *pc, int64_t pc_cstride, int32_t N, const double *T34, int32_t H, int32_t W, double fov_up,
                        double fov_down, int32_t *idx_ws, double *out, void *stream);
int efgh_sum_paint(const void *jobs_dev, int32_t njobs, void *stream);
int efgh_sum_colorize(const double *minmax, int64_t n, const uint8_t *lut, uint8_t *rgb, uint8_t *mask, void *stream);

/* ---- pose heads, forward (one launch each) -----------------------------------------------------------------------------------
 * efgh_pose_head_normal: softmax + L2 normalisation of the nd (2 or 3) "abs" logits, sign class = first argmax of the 2^nd sign
 *   logits decoded MSB-first, normal = abs * sign, rotation of the normal onto (dx,dy,dz) as a 4x4 (enet.py:161-176, hnet.py:59-77,
 *   torch_utils.py:105-146,170-200).  abs_out / normal: [B][nd], R44: [B][16].
 * efgh_pose_head_yaw: argmax of the n correlation scores -> yaw -> rotation onto e1 (fnet.py:87-91).
 * efgh_pose_cam_T_velo: A^-1 c_T A calib l_T (torch_utils.py:256-269); c_T rows of 3 with sample pitch ldc, out [B][12].       */
int efgh_pose_head_normal(const float *abs_logits, int64_t lda, const float *sgn_logits, int64_t lds, int32_t B, int32_t nd,
                          float dx, float dy, float dz, float *abs_out, float *normal, float *R44, void *stream);
int efgh_pose_head_yaw(const float *score, int64_t lds, int32_t B, int32_t n, float *R44, void *stream);
int efgh_pose_cam_T_velo(const float *c_T, int64_t ldc, const float *l_T, const float *calib, const float *A, int32_t B,
                         float *out34, void *stream);

/* ---- pose heads, training path: hand-written backward (forward-mode duals inside the kernel, one seed per input) ------------
 * efgh_pose_head_normal_bwd: d/d abs_logits [B][nd] of <g_abs, abs> + <g_normal, normal> + <g_R44, R>; any of the three incoming
 *   gradients may be NULL.  As in the reference the skew matrix of the rotation is detached (torch_utils.py:184,194): only the
 *   (1 - c)/s^2 factor carries a derivative; the sign logits get none (argmax).
 * efgh_pose_rotation_between: rotation of unit vectors src3 [B][3] onto (dx,dy,dz) (torch_utils.py:170-200), R44 [B][16].
 * efgh_pose_cam_T_velo_bwd: gradients of efgh_pose_cam_T_velo w.r.t. c_T ([B][9], may be NULL) and l_T ([B][16], may be NULL). */
int efgh_pose_head_normal_bwd(const float *abs_logits, int64_t lda, const float *sgn_logits, int64_t lds, int32_t B, int32_t nd,
                              float dx, float dy, float dz, const float *g_abs, const float *g_normal, const float *g_R44,
                              float *g_abs_logits, void *stream);
/* out = op(a) op(b), [B][4][4] each, op = transpose when the flag is set (fnet.py:101, gnet.py:180 and their gradients) */
int efgh_pose_mat44_mul(const float *a, const float *b, int32_t B, int32_t transpose_a, int32_t transpose_b, float *out, void *stream);
int efgh_pose_rotation_between(const float *src3, int32_t B, float dx, float dy, float dz, float *R44, void *stream);
int efgh_pose_cam_T_velo_bwd(const float *c_T, int64_t ldc, const float *l_T, const float *calib, const float *A,
                             const float *g_out34, int32_t B, float *g_cT33, float *g_lT44, void *stream);

/* ---- pose terms of the loss: E / H cosine + sign cross-entropy, F hard-negative-mined BCE, G translation smooth-L1, and the
 * ground truth they are measured against (losses/loss_utils.py:25-58, 77-144, 165-185, 227-262; losses/efghloss.py:19-38).
 * Predictions: e_gn_abs [B][3], e_gn_sgn [B][>=8], h_hrzn_abs [B][2], h_hrzn_sgn [B][>=4], f_score [B][W] (probabilities),
 * g_trs [B][3], e_l / f_l [B][16].  Ground truth: rand_init_l, rand_init_c ([B][3][3] or [B][4][4]), sensor2_T_sensor1 [B][16].
 * fwd:  gt72 [B][72] floats = e_gn 0..2 | e_l 3..18 | h_hrzn 19..21 | h_c (3x3) 22..30 | f_l 31..46 | g_trs 47..49 | g_l 50..65 |
 *       e_gn_abs 66..68 | h_hrzn_abs 69..70;  gt_cls2 [B][2] = sign classes (E, H);  gt_f_score / selected [B][W] (positives and
 *       positives + mined negatives);  partials: B * (7 + 2 * ceil(W / 256)) floats of scratch;  L11 = the eleven entries of the loss dictionary in `loss_name` order
 *       (total first; g_depth / g_mask from the scalars l_depth / l_mask of efgh_gimg_loss_fwd);  n_selected [1].
 * bwd:  gradients of <g_L11, L11> w.r.t. every prediction, and [2] w.r.t. (l_depth, l_mask).                                   */
typedef struct {
    const float *e_gn_abs, *e_gn_sgn, *h_hrzn_abs, *h_hrzn_sgn, *f_score, *g_trs, *e_l, *f_l;
    int64_t ld_e_gn_sgn, ld_h_hrzn_sgn, ld_f_score;
    const float *rand_init_l, *rand_init_c, *sensor2_T_sensor1;
    int32_t rand_init_l_dim, rand_init_c_dim;       /* 3 or 4: the perturbations as [B][3][3] rotations or [B][4][4] transforms */
    int32_t B, W, fov_pos_num;
    float fov_neg_ratio;
    float lambda_e_gn, lambda_h_hrzn, lambda_fov, lambda_g_trs, lambda_g_depth, lambda_g_mask;
} efgh_pose_loss_desc;
int efgh_pose_loss_fwd(const efgh_pose_loss_desc *d, const float *l_depth, const float *l_mask, float *gt72, int64_t *gt_cls2,
                       float *gt_f_score, float *selected, float *partials, float *L11, float *n_selected, void *stream);
int efgh_pose_loss_bwd(const efgh_pose_loss_desc *d, const float *g_L11, const float *selected, const float *n_selected,
                       float *g_e_gn_abs, float *g_e_gn_sgn, float *g_h_hrzn_abs, float *g_h_hrzn_sgn, float *g_f_score,
                       float *g_g_trs, float *g_e_l, float *g_ldepth_lmask, void *stream);

#ifdef __cplusplus
}
#endif
#endif
