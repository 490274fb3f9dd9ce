/* C ABI of libgeossl_hip.so — the MI355X (gfx950) hot path of GeoSSL's SchNet/PaiNN + DDM step.
 *
 * The reference (chao1224/GeoSSL) is pure Python: it has no FFI, the "interface" this library sits behind is
 * the set of ATen / torch_geometric / torch_scatter / torch_cluster calls its hot path makes.  Every entry
 * point below names the reference call site(s) it replaces (paths relative to the reference root).
 *
 * Conventions (all functions):
 *   - raw DEVICE pointers, explicit sizes, scalars by value, trailing hipStream_t;
 *   - return value is a hipError_t as int (0 = success); nothing throws, allocates or synchronises;
 *   - the caller owns every buffer, pre-allocates outputs / workspaces and keeps inputs alive until the
 *     stream reaches the call; re-entrant, no global state.
 *   - fp32 everywhere; "i64" index tensors are the reference's int64 tensors, internal indices are int32.
 */
#ifndef GEOSSL_HIP_H
#define GEOSSL_HIP_H
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else
typedef struct ihipStream_t* hipStream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GEOSSL_ABI_VERSION 1
#define GEOSSL_MAX_L 12 /* max interaction blocks handled by the batched-by-layer kernels */
#define GEOSSL_TN_MAX 32 /* max problems in one batched weight-gradient launch */
#define GEOSSL_LOSS_PARTIALS 256 /* block partials of the per-row loss that geossl_ddm_loss_fwd leaves in its workspace */

/* epilogue flags of geossl_linear */
#define GEOSSL_EPI_BIAS 1      /* + bias[n] */
#define GEOSSL_EPI_SSP 2       /* ShiftedSoftplus (schnet.py:210-216) */
#define GEOSSL_EPI_RESIDUAL 4  /* + res[r][n]   (h = h + block(h), schnet.py:97) */
#define GEOSSL_EPI_MUL_DSSP 8  /* * d ssp/dx recovered from the saved ssp OUTPUT tprev[r][n] (backward) */
#define GEOSSL_CHAIN_NEW_INPUT 32  /* geossl_linear_chain, F = 128: the stage reads its own input `xin` (a Linear over a
                                      wide input = several F-wide passes in one launch) */
#define GEOSSL_CHAIN_ADD_PREV 64   /* ... and adds the result of the stage before it (kept in registers) */
#define GEOSSL_EPI_SILU 128       /* geossl_linear_chain, F = 128: Y = X W^T + b goes to `out` as it is (the backward needs the
                                    pre-activation), silu(Y) to `out_act` (may be NULL) and into the next stage -
                                    Dense(F, F, silu) + the Dense behind it, painn_utils.py:27-35 */
#define GEOSSL_EPI_MUL_DSILU 256  /* ... with tprev: * silu'(tprev), tprev = the saved PRE-activation (backward) */
#define GEOSSL_CHAIN_SAME_INPUT 16 /* geossl_linear_chain, F = 128: the stage takes the input of the stage before it
                                      (several F -> F blocks of one wide Linear in one launch) instead of its result */

int geossl_abi_version(void);

/* ---- batch layout (position independent; built once per collated batch) -------------------------------
 * Replaces the index bookkeeping of BatchAtomTuple.from_data_list (Geom3D/dataloaders/dataloaders_AtomTuple.py:57-73)
 * and the `batch[-1].item()+1` of num_graphs (:75-78).  `batch` must be sorted ascending (collated batches are).
 *   mol_ptr[B+1]  : atom offsets; pair_ptr[B+1] : offsets of the n(n-1)/2 "pair slots" (i<j, lexicographic —
 *                   the same enumeration as AtomTupleExtractor "combination", :22-23)
 *   stats[4]      : {max atoms per molecule, total pair slots P, 1 if batch is not sorted / out of range, 0}
 *   pair_i/pair_j : atom ids (global) of every pair slot                                                   */
int geossl_layout_build(const int64_t* batch, int64_t N, int64_t B, int32_t* mol_ptr, int32_t* pair_ptr,
                        int64_t* stats, hipStream_t stream);
/* Device-side AtomTupleExtractor (dataloaders_AtomTuple.py:15-37 with ratio = 1) including the node offset of the
 * collate (:64-65): super_edge_index rows out0 / out1.  option 0 = "combination" (i<j, itertools.combinations
 * order), 1 = "permutation" (i != j, itertools.permutations order).  tuple_ptr[B+1] (int64): exclusive prefix sum
 * of the per-molecule tuple counts n(n-1)/2 or n(n-1).                                                        */
int geossl_atom_tuples(const int32_t* mol_ptr, const int64_t* tuple_ptr, int64_t B, int option, int64_t* out0,
                       int64_t* out1, hipStream_t stream);
int geossl_pair_index_fill(const int32_t* mol_ptr, const int32_t* pair_ptr, int64_t B, int32_t* pair_i,
                           int32_t* pair_j, hipStream_t stream);

/* ---- radius graph (K1) — torch_cluster.radius_graph via torch_geometric (schnet.py:91;
 * datasets_3D_Radius.py:120) + the edge length of schnet.py:93.
 * Semantics: per molecule, per target i scan sources j ascending, hit when fl32(dx^2+dy^2+dz^2) < r2 (strict,
 * no FMA), stop after `cap` hits (cap = max_num_neighbors + 1, the self hit counts), drop j == i.
 *   _count : deg[N] = in-degree of every target
 *   _fill  : edge_ptr[N+1] = exclusive scan of deg (caller); writes edge_index as two int64 rows
 *            (src = source j, dst = target i; target-major, sources ascending) and edge_weight[E]
 *   geossl_pair_geometry : the same graph in pair-slot form for the fused SchNet path:
 *            pair_d[P] = |x_i - x_j|, pair_c[P] = 0.5*(cos(d*pi/cutoff)+1) (the CFConv envelope, schnet.py:186),
 *            pair_flag[P] bit0 = edge (j -> i) present, bit1 = edge (i -> j) present, for slot (i<j).       */
int geossl_radius_graph_count(const float* pos, const int32_t* mol_ptr, int64_t B, int max_n, float r2, int cap,
                              int32_t* deg, hipStream_t stream);
int geossl_radius_graph_fill(const float* pos, const int32_t* mol_ptr, int64_t B, int max_n, float r2, int cap,
                             const int64_t* edge_ptr, int64_t* edge_src, int64_t* edge_dst, float* edge_weight,
                             hipStream_t stream);
int geossl_pair_geometry(const float* pos, const int32_t* mol_ptr, const int32_t* pair_ptr, int64_t B, int max_n,
                         float r2, int cap, float cutoff, float* pair_d, float* pair_c, uint8_t* pair_flag,
                         hipStream_t stream);

/* ---- Gaussian smearing (K2) — GaussianSmearing.forward, schnet.py:205-207: out[e][g] = exp(coeff*(d-off_g)^2) */
int geossl_rbf_fwd(const float* d, int64_t E, const float* offset, int G, float coeff, float* out,
                   hipStream_t stream);

/* ---- ShiftedSoftplus as a stand-alone op — ShiftedSoftplus.forward, schnet.py:210-216 (the module is usable on its
 * own in the reference; inside the hot path it is fused into the GEMM epilogues):  y = softplus(x) - log 2;
 * backward from the saved OUTPUT: dx = dy * sigmoid(x) = dy * (1 - 0.5 exp(-y)).                                 */
int geossl_ssp_fwd(const float* x, int64_t n, float* y, hipStream_t stream);
int geossl_ssp_bwd(const float* y, const float* dy, int64_t n, float* dx, hipStream_t stream);

/* ---- continuous-filter network for all interaction blocks (K3) — InteractionBlock.mlp applied in
 * CFConv.forward, schnet.py:141-145,186-187:  Wf_l[p] = (ssp(rbf(d_p) A1_l^T + b1_l) A2_l^T + b2_l) * C(d_p),
 * C(d) = 0.5*(cos(d*pi/cutoff)+1).  One launch covers every layer l < L and every pair slot p < P.
 *   T (optional, training) : saved hidden activation ssp(.) [L][P][F];  Wf : [L][P][F]                      */
typedef struct {
  const float* w1[GEOSSL_MAX_L]; /* mlp.0.weight [F][G] */
  const float* b1[GEOSSL_MAX_L]; /* mlp.0.bias   [F]    */
  const float* w2[GEOSSL_MAX_L]; /* mlp.2.weight [F][F] */
  const float* b2[GEOSSL_MAX_L]; /* mlp.2.bias   [F]    */
} GeosslFilterWeights;
int geossl_cfconv_filter_fwd(const float* pair_d, const float* pair_c, int64_t P, const GeosslFilterWeights* w, int L,
                             int F, int G, const float* offset, float coeff, float* T, float* Wf, hipStream_t stream);

/* Backward of K3 with respect to the filter-network weights, all blocks at once (positions carry no gradient on
 * the DDM path).  The upstream gradient dWf[p] = flag0*dagg[i]*x[j] + flag1*dagg[j]*x[i] is never materialised: it is
 * rebuilt from the per-layer atom tensors x_l = conv.lin1(h) and dagg_l = dL/d(aggregate) (both [N][F]) staged in LDS.
 *   dA2_l = sum_p dO^T T,  db2_l = sum_p dO,  dU = ((dO) A2_l) * ssp'(.),  dA1_l = sum_p dU^T rbf(d),  db1_l = sum_p dU
 * with dO = dWf * C(d).  Deterministic (per-wave / per-block partials + fixed-order reduction).                 */
typedef struct {
  const float* x[GEOSSL_MAX_L];    /* [N][F] */
  const float* dagg[GEOSSL_MAX_L]; /* [N][F] */
} GeosslFilterGradIn;
typedef struct {
  float* dw1[GEOSSL_MAX_L];
  float* db1[GEOSSL_MAX_L];
  float* dw2[GEOSSL_MAX_L];
  float* db2[GEOSSL_MAX_L];
} GeosslFilterGradOut;
int64_t geossl_cfconv_filter_bwd_workspace_floats(int64_t P, int L, int F, int G);
int geossl_cfconv_filter_bwd(const float* pair_d, const float* pair_c, const uint8_t* pair_flag, const int32_t* pair_i,
                             const int32_t* pair_j, int64_t P, int64_t N, const GeosslFilterWeights* w,
                             const GeosslFilterGradIn* g, int L, int F, int G, const float* offset, float coeff,
                             const float* T, const GeosslFilterGradOut* out, float* workspace, int accumulate,
                             hipStream_t stream);

/* ---- position gradient through K1-K4 (first order) — the d/d pos of finetune_md17.py:46 through
 * schnet.py:91-93 (edge length), :186-187 (cosine envelope) and :205-207 (Gaussian smearing); SURVEY 8(f) N3.
 * Same inputs as geossl_cfconv_filter_bwd plus the filter rows Wf; dd[l][p] = dL/dd_p contributed by block l:
 *   sum_c (flag0*dagg[i][c]*x[j][c] + flag1*dagg[j][c]*x[i][c]) * dWf[p][c]/dd,
 *   dWf/dd = C(d) * W2 (ssp'(.) * (W1 rbf'(d)))  +  C'(d)/C(d) * Wf        (the filter row differentiated forward).
 * geossl_pair_position_grad sums dd over l and applies d|pos_a - pos_b|/d pos to both atoms of every pair slot:
 *   dpos[a] = sum_b (sum_l dd[l][slot(a,b)]) * (pos[a] - pos[b]) / d(a,b)        (fixed order, no atomics).   */
int geossl_cfconv_filter_dpos(const float* pair_d, const float* pair_c, const uint8_t* pair_flag, const int32_t* pair_i,
                              const int32_t* pair_j, int64_t P, const GeosslFilterWeights* w,
                              const GeosslFilterGradIn* g, int L, int F, int G, const float* offset, float coeff,
                              float cutoff, const float* T, const float* Wf, float* dd, hipStream_t stream);
int geossl_pair_position_grad(const float* pos, const float* pair_d, const float* dd, const int32_t* mol_ptr,
                              const int32_t* pair_ptr, int64_t B, int64_t P, int L, float* dpos, hipStream_t stream);

/* ---- neighbour aggregation (K4) — MessagePassing.propagate(aggr="add") with message x_j * W
 * (schnet.py:190,194-195): out[i] = sum over edges (j -> i), j ascending, of x[j] * Wf[slot(i,j)].
 * swap = 1 runs the transposed graph (backward w.r.t. x): out[j] = sum over edges (j -> i) of x[i]*Wf.
 * order (may be NULL): a permutation of the molecules, the sequence in which they are started (largest first for
 * ragged batches; it does not change any result).                                                              */
int geossl_cfconv_aggregate(const float* x, const float* Wf, const uint8_t* pair_flag, const int32_t* mol_ptr,
                            const int32_t* pair_ptr, const int32_t* order, int64_t B, int max_n, int F, int swap,
                            float* out, hipStream_t stream);
/* The same aggregation from a host-built work list (ragged batches): work entries molecule | part << 24 (molecule < 2^24,
 * part 0 .. 254; -1 = padding) in EIGHT queues of nwork / 8 entries each, one per XCD (workgroup b takes entry b / 8 of
 * queue b mod 8: the items of one molecule, which read each other's filter rows, share an L2), largest molecules first
 * within a queue.  A molecule of n atoms has geossl_aggregate_parts(n) entries
 * (1, 2 or 4 up to 33 atoms: the 27..33-atom molecules are shared by that many waves, one group of target atoms each; n
 * above 33 atoms - Molecule3D with hydrogens, datasets_Molecule3D.py:65 - where a wave sums ONE target atom without a
 * size class) - every sum is still formed by one wave in the order of geossl_cfconv_aggregate, bit for bit.
 * max_n <= 255, 32 < F <= 128.                                                                                      */
int geossl_aggregate_parts(int n);
/* ... and with ONE work item per target atom for every molecule (work[i] = molecule | target << 24, all atoms of the
 * launch): small batches, where a launch is bound by its longest walk (a chain of memory round trips) and not by
 * bytes - every filter row is read by both of its atoms.  Same sums, bit for bit.                                      */
int geossl_cfconv_aggregate_targets_dyn(const float* x, const float* Wf, const uint8_t* pair_flag, const int32_t* mol_ptr,
                                        const int32_t* pair_ptr, const int32_t* work, int64_t nwork, int max_n, int F,
                                        int swap, float* out, const int32_t* dyn_nwork, hipStream_t stream);
int geossl_cfconv_aggregate_work(const float* x, const float* Wf, const uint8_t* pair_flag, const int32_t* mol_ptr,
                                 const int32_t* pair_ptr, const int32_t* work, int64_t nwork, int max_n, int F,
                                 int swap, float* out, hipStream_t stream);

/* Gradient of geossl_cfconv_aggregate with respect to the filter rows, as a tensor:
 * out[p][c] = f0 a[i][c] b[j][c] + f1 a[j][c] b[i][c] for pair slot p = (i < j) with edge flags f0 (j -> i), f1 (i -> j)
 * (exchanged when swap = 1).  The first-order path never materialises it (geossl_cfconv_filter_bwd); the second-order
 * path (training on forces, finetune_md17.py:46-54) keeps it differentiable.                                      */
int geossl_pair_product(const float* a, const float* b, const int32_t* pair_i, const int32_t* pair_j,
                        const uint8_t* pair_flag, int64_t P, int F, int swap, float* out, hipStream_t stream);

/* ---- atom-row Linear — ATen Linear at schnet.py:99,101,166,189,191 and its autograd.
 * Y[r][n] = epi(sum_k X[r][k] * Bm[k][n]); transB=1: W is torch layout [NO][K] (forward);
 * transB=0: W is [K][NO] (backward w.r.t. input: dX = dY W).  K % 8 == 0, K,NO <= 256.  ldx / ldy: row strides of
 * X and of Y (res and tprev share ldy), so column slices of wider tensors can be read / written in place.   */
int geossl_linear(const float* X, int ldx, const float* W, const float* bias, const float* res, const float* tprev,
                  float* Y, int ldy, int64_t R, int K, int NO, int transB, int flags, hipStream_t stream);

/* Prepared weights.  geossl_linear re-shapes W into its MFMA operand image (bf16 pieces in fragment order) in every
 * block of every launch; a weight that is used by several launches between two optimiser steps (forward, backward,
 * both views) can be converted once instead: geossl_linear_prepare builds the images of up to GEOSSL_PREPARE_MAX
 * weights of one shape in a single launch, geossl_linear_prepared is geossl_linear reading such an image
 * (same arithmetic, bit-identical results).  geossl_linear_image_words: size of one image in 32-bit words, 0 if the
 * shape has no prepared path (K not in {32, 64, 128}).                                                       */
#define GEOSSL_PREPARE_MAX 64 /* weights in one geossl_linear_prepare / geossl_chain_prepare launch */
typedef struct {
  const float* W[GEOSSL_PREPARE_MAX];
  uint32_t* image[GEOSSL_PREPARE_MAX];
  int ldw[GEOSSL_PREPARE_MAX]; /* row stride of W in floats (a multiple of 4), 0 = dense; geossl_chain_prepare only: a column
                             block of a wider weight (PaiNN's Dense(2F, F), painn.py:87) is converted where it lies */
  int tb[GEOSSL_PREPARE_MAX];  /* geossl_chain_prepare only: 0 = the call's transB, 1 = transB 1, 2 = transB 0 for this weight
                                  (the forward and the backward image of a weight from one launch) */
} GeosslPrepareBatch;
int64_t geossl_linear_image_words(int K, int NO);
int geossl_linear_prepare(const GeosslPrepareBatch* batch, int nprob, int K, int NO, int transB, hipStream_t stream);
int geossl_linear_prepared(const float* X, int ldx, const uint32_t* image, const float* bias, const float* res,
                           const float* tprev, float* Y, int ldy, int64_t R, int K, int NO, int flags,
                           hipStream_t stream);

/* Chains of square atom-row Linear layers in one launch.  Between two neighbour aggregations the reference applies
 * three row-local layers back to back - conv.lin2 (schnet.py:191), act + lin + residual (:165-166,97), the next
 * block's conv.lin1 (:189); after the last block lin2, lin and the head (:99-101) - and autograd walks the same chains
 * backwards.  Stage s: Y_s = epi_s(X_s W_s^T + b_s), X_{s+1} = Y_s, with geossl_linear's epilogue (flags:
 * GEOSSL_EPI_SSP; tprev != NULL: * ssp'(tprev); res != NULL: + res) and Y_s stored to `out` when it is not NULL
 * (row stride ld, shared with res / tprev).  Every stage is F -> F, F in {32, 64, 128}.  `image`: operand image of the
 * stage's weight from geossl_chain_prepare (transB as in geossl_linear: 1 = W is [NO][K] (forward), 0 = W is [K][NO]
 * (dX = dY W)), geossl_chain_image_words(F) 32-bit words each.  The image is opaque: MFMA operand fragments of W split
 * into 16-bit pieces (F = 128: two fp16 pieces of W scaled by a power of two per 32-column output block, the four
 * exponents stored behind the fragments; F = 64 / 32: three bf16 pieces); results carry fp32-GEMM accuracy.        */
#define GEOSSL_CHAIN_MAX 5 /* F = 128; the F = 64 / 32 forms take up to 3 stages */
typedef struct {
  const uint32_t* image;
  const float* bias;  /* may be NULL */
  const float* res;   /* may be NULL */
  const float* tprev; /* may be NULL */
  float* out;         /* may be NULL: the stage's result is only consumed by the next stage */
  int ld;
  int flags;
  const float* xin;   /* GEOSSL_CHAIN_NEW_INPUT: the stage's own input rows [R][F] (row stride ldxin), else NULL */
  int ldxin;
  int pad_;
  float* out_act;     /* GEOSSL_EPI_SILU: silu of the stage's result (row stride ld), may be NULL */
} GeosslChainStage;
typedef struct {
  int nstage;
  GeosslChainStage st[GEOSSL_CHAIN_MAX];
} GeosslChain;
int64_t geossl_chain_image_words(int F);
int geossl_chain_prepare(const GeosslPrepareBatch* batch, int nprob, int F, int transB, hipStream_t stream);
int geossl_linear_chain(const float* X, int ldx, const GeosslChain* chain, int64_t R, int F, hipStream_t stream);

/* The layer loop of the SchNet backbone in one launch (schnet.py:95-101 between the filter network and the heads, and
 * its backward): a list of operations that each block of the launch applies, in order, to the rows of the molecules it
 * owns - kind 0: the chain of geossl_linear_chain over those rows (plain stages: dense [N][F] operands, at most three),
 * kind 1: geossl_cfconv_aggregate for those molecules (x = X, filter rows Wf, destination out; swap as there).  `plan`
 * [nblocks][4] int32 = {first row, end row, first molecule, end molecule} of a block (at most 96 rows; consecutive
 * blocks cover the batch).  uniform != 0: every molecule has max_n atoms.  stagger: the second half of the grid starts
 * that many sleeps late (0 = in step; for experiments).  F = 128 and uniform batches of at most 20 atoms per molecule;
 * anything else returns hipErrorInvalidValue and the caller launches the operations one by one.  Same arithmetic as
 * the separate launches: results bit-identical. */
#define GEOSSL_LOOP_MAX_OPS 14
typedef struct {
  int kind; /* 0 = chain, 1 = aggregation */
  int swap;
  const float* X;
  const float* Wf;
  float* out;
  GeosslChain chain;
} GeosslLoopOp;
int geossl_schnet_layer_loop(const GeosslLoopOp* ops, int nops, const int32_t* plan, int nblocks, const int32_t* mol_ptr,
                             const int32_t* pair_ptr, const uint8_t* pair_flag, int max_n, int uniform, int64_t N, int F,
                             int stagger, hipStream_t stream);
/* The same loop over RAGGED molecules (1 .. 33 atoms each) of a SMALL batch: block b owns `mols_per_block` (1 or 2)
 * consecutive molecules, found from mol_ptr on the device - no host plan, so the launch also serves a capacity bucket
 * (the `_dyn` section below), whose index structures are device data; every wave takes the unrolled walk of its
 * molecule's size class.  B molecules in at most 1024 blocks.  N = rows of the atom tensors (a capacity is fine: no row
 * past mol_ptr[B] is touched).  Results bit-identical to the separate launches. */
int geossl_schnet_layer_loop_ragged(const GeosslLoopOp* ops, int nops, const int32_t* mol_ptr, const int32_t* pair_ptr,
                                    const uint8_t* pair_flag, int64_t B, int mols_per_block, int64_t N, int F,
                                    hipStream_t stream);

/* batched weight gradients: dW_z[m][n] (+)= sum_r A_z[r][m]*B_z[r][n], db_z[m] (+)= sum_r A_z[r][m];
 * lda / ldb / ldw: row strides of A_z, B_z, dW_z (M, N <= 128 per problem: wider layers are tiled by the caller)  */
typedef struct {
  const float* A[GEOSSL_TN_MAX];
  const float* B[GEOSSL_TN_MAX];
  float* dW[GEOSSL_TN_MAX];
  float* db[GEOSSL_TN_MAX]; /* may be NULL */
} GeosslTnBatch;
typedef struct {
  float* out[GEOSSL_TN_MAX];
} GeosslReduceBatch;
void geossl_tn_plan(int64_t R, int nprob, int* chunk, int* nblk);
int64_t geossl_tn_workspace_floats(int64_t R, int M, int N, int nprob);
int geossl_linear_wgrad(const GeosslTnBatch* batch, int nprob, int64_t R, int M, int N, int lda, int ldb, int ldw,
                        float* workspace, int accumulate, hipStream_t stream);

/* ---- embedding — torch.nn.Embedding at schnet.py:89 (z is a strided int64 view x[:,0])                     */
int geossl_embedding_fwd(const int64_t* z, int64_t z_stride, const float* table, int num_classes, int64_t N, int F,
                         float* out, int32_t* status, hipStream_t stream);
int64_t geossl_embedding_bwd_workspace_floats(int num_classes, int F);
int geossl_embedding_bwd(const int64_t* z, int64_t z_stride, const float* dh, int num_classes, int64_t N, int F,
                         float* dtable, float* workspace, int accumulate, hipStream_t stream);

/* ---- readout — torch_scatter.scatter(h, batch, dim=0, reduce) at schnet.py:115 / painn.py:266
 * mean = sum / max(count,1).  _bwd: dh[a] = dout[batch[a]] (/count).                                          */
int geossl_segment_reduce_fwd(const float* h, const int32_t* mol_ptr, int64_t B, int F, int mean, float* out,
                              hipStream_t stream);
/* F.normalize(h, dim=-1) of the optional --normalize branch of do_DDM (pretrain_GeoSSL.py:193-195):
 * y = h / max(||h||_2, eps) per row; norm[N] (may be NULL in inference) is kept for the backward
 * dh = (g - y (g . y)) / max(||h||, eps).                                                                     */
int geossl_row_normalize_fwd(const float* h, int64_t N, int F, float eps, float* y, float* norm, hipStream_t stream);
int geossl_row_normalize_bwd(const float* g, const float* y, const float* norm, int64_t N, int F, float eps,
                             float* dh, hipStream_t stream);
int geossl_segment_reduce_bwd(const float* dout, const int32_t* mol_ptr, int64_t B, int F, int mean, float* dh,
                              int accumulate, hipStream_t stream);

/* ---- DDM pieces — examples/pretrain_GeoSSL.py:68-74,199-205 and examples/NCSN.py:183-220 ----------------- */
/* perturb (:72): out = pos + noise */
int geossl_axpy(const float* a, const float* b, float alpha, int64_t n, float* out, hipStream_t stream);
/* super-edge length (:199-205): out[s] = sqrt(sum((pos[u]-pos[v])^2)) */
int geossl_pair_distance(const float* pos, const int64_t* sei0, const int64_t* sei1, int64_t S, float* out,
                         hipStream_t stream);
/* batch.to(device)-side plumbing of a replayed step (:248): two device-to-device copies (atom types and positions of
 * the next batch into a captured graph's input buffers) as one launch; byte counts, multiples of 4 */
int geossl_copy2(void* dst0, const void* src0, int64_t bytes0, void* dst1, const void* src1, int64_t bytes1,
                 hipStream_t stream);
/* The five random draws of a DDM step (perturb :72: N(mu, sigma) per coordinate, n_pos = 3 N values; per head a noise
 * level in [0, K) per molecule NCSN.py:190 and N(0, 1) per super-edge :194) in one launch, for a caller that owns its
 * random stream (DDMTrainer with device noise): Philox4x32-10 keyed by the 64-bit *seed on the device.  The reference's
 * own loop keeps torch's calls (do_DDM: same generator, same values). */
int geossl_ddm_noise(const int64_t* seed, float mu, float sigma, int64_t n_pos, int64_t S, int64_t B, int K1, int K2,
                     float* pos_noise, int64_t* noise_level_1, float* distance_noise_1, int64_t* noise_level_2,
                     float* distance_noise_2, hipStream_t stream);
/* the same with the 64-bit Philox key passed by value (a caller that derives it on the host - from torch's generator
 * state - saves the launch that draws it on the device) */
int geossl_ddm_noise_seeded(uint64_t seed, float mu, float sigma, int64_t n_pos, int64_t S, int64_t B, int K1, int K2,
                            float* pos_noise, int64_t* noise_level_1, float* distance_noise_1, int64_t* noise_level_2,
                            float* distance_noise_2, hipStream_t stream);
/* both views at once (:68-74 and :199-205 for a fused two-view batch): pos2 [2N][3] = [pos ; pos + noise], d01 / d02 [S] =
 * super-edge lengths in the clean / perturbed view - geossl_axpy, the concatenation and two geossl_pair_distance calls
 * in one launch, same arithmetic; z2 != NULL: also z2 [2N] = the atom types z[i * z_stride] of the N atoms, twice */
int geossl_ddm_views(const float* pos, const float* noise, const int64_t* sei0, const int64_t* sei1, int64_t N, int64_t S,
                     float* pos2, float* d01, float* d02, const int64_t* z, int64_t z_stride, int64_t* z2,
                     hipStream_t stream);
/* per-batch bookkeeping for the NCSN head: se_ptr[B+1] = first super-edge of every molecule (needs
 * batch[sei0] non-decreasing and both ends in one molecule: true for collated batches),
 * stats = {max(edge2graph)+1 (the divisor of NCSN.py:212), 1 if the ordering assumption fails}; and the
 * atom -> incident (super-)edge lists (inc_ptr from an exclusive scan of inc_cnt by the caller); sides bit0 /
 * bit1 select matches on row 0 / row 1 of the index (3 = both, the NCSN case; 1 or 2 = PaiNN's per-target /
 * per-source edge lists).                                                                                     */
int geossl_super_edge_ptr(const int64_t* batch, const int64_t* sei0, const int64_t* sei1, int64_t S, int64_t B,
                          int32_t* se_ptr, int64_t* stats, hipStream_t stream);
int geossl_incidence_count(const int64_t* batch, const int64_t* sei0, const int64_t* sei1, const int32_t* se_ptr,
                           int64_t N, int sides, int32_t* inc_cnt, hipStream_t stream);
int geossl_incidence_fill(const int64_t* batch, const int64_t* sei0, const int64_t* sei1, const int32_t* se_ptr,
                          int64_t N, int sides, const int64_t* inc_ptr, int32_t* inc_idx, hipStream_t stream);

typedef struct {
  const float* in_w1; /* input_distance_mlp.layers.0.weight [F][1] */
  const float* in_b1; /* [F] */
  const float* in_w2; /* input_distance_mlp.layers.1.weight [1][F] */
  const float* in_b2; /* [1] */
  const float* o1_w;  /* output_mlp.layers.0.weight [F][F+1] */
  const float* o1_b;  /* [F] */
  const float* o2_w;  /* output_mlp.layers.1.weight [F/2][F] */
  const float* o2_b;  /* [F/2] */
  const float* o3_w;  /* output_mlp.layers.2.weight [1][F/2] */
  const float* o3_b;  /* [1] */
  const float* sigmas; /* [K] */
} GeosslNcsnWeights;
typedef struct {
  float* in_w1; float* in_b1; float* in_w2; float* in_b2;
  float* o1_w; float* o1_b; float* o2_w; float* o2_b; float* o3_w; float* o3_b;
} GeosslNcsnGrads;
/* Saved per-row state of the head (training): a1 [S][F], a2 [S][F/2] (post-relu), pd[S] (perturbed distance),
 * emb[S], gscale[S] = d loss_e / d(output_mlp out).                                                         */
typedef struct {
  float* a1; float* a2; float* pd; float* emb; float* gscale;
} GeosslNcsnSaved;
/* K5 forward: loss_e[S] (NCSN.py:209) from node features h [N][F], distances d[S], the two random draws
 * noise_level[B] (i64, :190) and distance_noise[S] (:194) given as inputs.                                   */
/* workspace (GEOSSL_LOSS_PARTIALS floats, may be NULL): receives one partial sum of loss_e per block of the launch, zeros
 * behind them - geossl_loss_reduce_partials finishes the sum without another pass over loss_e */
int64_t geossl_ddm_loss_fwd_workspace_floats(int F);
int geossl_ddm_loss_fwd(const float* h, const int64_t* batch, const int64_t* sei0, const int64_t* sei1, int64_t S,
                        const float* distance, const int64_t* noise_level, const float* distance_noise,
                        const GeosslNcsnWeights* w, int F, float anneal_power, float* loss_e,
                        const GeosslNcsnSaved* saved, float* workspace, hipStream_t stream);
/* loss = sum_s loss_e[s] / divisor (NCSN.py:210-212), fixed-order two-stage sum; out_scale multiplies the result
 * (0.5 for the (l1+l2)/2 of pretrain_GeoSSL.py:210) and accumulate adds to *loss.                            */
int64_t geossl_loss_reduce_workspace_floats(int64_t S);
int geossl_loss_reduce(const float* loss_e, int64_t S, const int64_t* stats_divisor, float out_scale, float* loss,
                       float* workspace, int accumulate, hipStream_t stream);
/* The two heads of a DDM step (pretrain_GeoSSL.py:207-208: same super-edges, each head its own view, weights and noise)
 * in the SAME launches - at the reference's batch size one head's row pass fills a quarter of the chip.
 * geossl_ddm_loss_fwd2: both row passes (workspace as in geossl_ddm_loss_fwd); geossl_loss_reduce_partials2: loss =
 * scale0 * mean_0 + scale1 * mean_1 from the two workspaces; geossl_ddm_loss_bwd_fused2: both one-pass backwards, their
 * reductions, and dh of both heads (geossl_incidence_gather) - 3 launches where the single-head calls make 8. */
typedef struct {
  const float* h;              /* [N][F] node features of the head's view */
  const float* distance;       /* [S] */
  const int64_t* noise_level;  /* [B] */
  const float* distance_noise; /* [S] */
  GeosslNcsnWeights w;
  GeosslNcsnSaved saved;       /* all NULL: nothing kept for a backward */
  float anneal_power;
  float pad_;
  float* loss_e;               /* [S] */
  float* workspace;            /* GEOSSL_LOSS_PARTIALS floats */
} GeosslNcsnHeadFwd;
typedef struct {
  const float* h;
  GeosslNcsnWeights w;
  GeosslNcsnSaved saved;
  float out_scale;
  float pad_;
  float* dfeat;                /* [S][F] */
  float* demb;                 /* [S] */
  float* grow;                 /* [S] */
  GeosslNcsnGrads grads;
  float* workspace;            /* geossl_ddm_loss_bwd_fused_workspace_floats(S, F) */
  float* dh;                   /* [N][F], may be NULL */
} GeosslNcsnHeadBwd;
int geossl_ddm_loss_fwd2(const GeosslNcsnHeadFwd* heads, const int64_t* batch, const int64_t* sei0, const int64_t* sei1,
                         int64_t S, int F, hipStream_t stream);
int geossl_loss_reduce_partials2(const float* partial0, const float* partial1, const int64_t* stats_divisor, float scale0,
                                 float scale1, float* loss, hipStream_t stream);
int geossl_ddm_loss_bwd_fused2(const GeosslNcsnHeadBwd* heads, const int64_t* sei0, const int64_t* sei1, int64_t S,
                               int64_t N, int F, const int64_t* stats_divisor, const float* gout, const int64_t* inc_ptr,
                               const int32_t* inc_idx, int accumulate, hipStream_t stream);
/* the same loss from the block partials geossl_ddm_loss_fwd left in its workspace (fixed order: blocks in sequence) */
int geossl_loss_reduce_partials(const float* partial, const int64_t* stats_divisor, float out_scale, float* loss,
                                int accumulate, hipStream_t stream);
/* K5 backward.  gout = upstream gradient of the head's scalar loss (device scalar, may be NULL = 1),
 * row pass: dz1 [S][F] (grad at output_mlp hidden 1 pre-activation), dfeat [S][F] (grad w.r.t. h_u + h_v),
 * demb[S]; then weight gradients; then dh[a] (+)= sum of dfeat over incident super-edges (fixed order).      */
int geossl_ddm_loss_bwd_rows(const GeosslNcsnWeights* w, const GeosslNcsnSaved* saved, int64_t S, int F,
                             const int64_t* stats_divisor, float out_scale, const float* gout, float* dz1,
                             float* dfeat, float* demb, float* grow, hipStream_t stream);
int64_t geossl_ddm_loss_bwd_workspace_floats(int64_t S, int F);
int geossl_ddm_loss_bwd_weights(const float* h, const int64_t* sei0, const int64_t* sei1, int64_t S, int F,
                                const GeosslNcsnWeights* w, const GeosslNcsnSaved* saved, const float* dz1,
                                const float* demb, const float* grow, const GeosslNcsnGrads* grads,
                                float* workspace, int accumulate, hipStream_t stream);
/* The same backward in one pass over the rows (ncsn_bwd.hip): row gradients dfeat / demb / grow AND every weight
 * gradient of the head (all members of `grads`), nothing per-row re-read and no dz1 in HBM.  h [N][F] is the head's
 * input (its rows are gathered again for the layers.0 weight gradient).                                        */
int64_t geossl_ddm_loss_bwd_fused_workspace_floats(int64_t S, int F);
int geossl_ddm_loss_bwd_fused(const float* h, const int64_t* sei0, const int64_t* sei1, int64_t S, int64_t N, int F,
                              const GeosslNcsnWeights* w, const GeosslNcsnSaved* saved, const int64_t* stats_divisor,
                              float out_scale, const float* gout, float* dfeat, float* demb, float* grow,
                              const GeosslNcsnGrads* grads, float* workspace, int accumulate, hipStream_t stream);
int geossl_incidence_gather(const float* dfeat, const int64_t* inc_ptr, const int32_t* inc_idx, int64_t N, int F,
                            float* dh, int accumulate, hipStream_t stream);

/* ---- PaiNN (BASELINE config 5) — Geom3D/models/painn.py:32-66,91-114,216-269 -------------------------------
 * idx_i / idx_j are rows 0 / 1 of radius_edge_index (int64).  Edge geometry (painn.py:232-239): dir[E][3] = r_ij/d,
 * fcut[E] = cosine cutoff * [d < cutoff] (painn_utils.py:152-154), phi[E][R] = Gaussian RBF (painn_utils.py:99-103).
 * interaction_fwd (painn.py:54-64): inc_ptr/inc_idx = edges grouped by idx_i (geossl_incidence_*, sides = 1);
 * xc = interatomic_context_net(q) [N][3F]; Wf/bf = the layer's 3F rows of filter_net; q [N][F], mu [N][3][F].
 * interaction_bwd: edges grouped by idx_j (sides = 2); returns d xc, d mu (incl. the residual) and the
 * filter_net gradient rows of the layer.  mix_* are the element-wise parts of PaiNNMixing (painn.py:100-113):
 * mm = mu_channel_mix(mu) [N][3][2F], ctx = [q, |mu_V|] [N][2F], dot = sum_xyz mu_V*mu_W, xx = context net out.  */
int geossl_painn_edge_geom(const float* pos, const int64_t* idx_i, const int64_t* idx_j, int64_t E, float cutoff,
                           const float* offsets, const float* widths, int R, float* dir, float* fcut, float* phi,
                           hipStream_t stream);
int geossl_silu_fwd(const float* u, int64_t n, float* y, hipStream_t stream);
int geossl_silu_bwd(const float* u, const float* dy, int64_t n, float* du, hipStream_t stream);
int geossl_painn_interaction_fwd(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                 const int64_t* inc_ptr, const int32_t* inc_idx, const float* phi, const float* fcut,
                                 const float* dir, const float* Wf, const float* bf, int64_t N, int F, int R,
                                 float* q_out, float* mu_out, hipStream_t stream);
int64_t geossl_painn_interaction_bwd_workspace_floats(int64_t N, int F, int R);
int geossl_painn_interaction_bwd(const float* dq_out, const float* dmu_out, const float* mu, const float* xc,
                                 const int64_t* idx_i, const int64_t* inc_ptr, const int32_t* inc_idx, const float* phi,
                                 const float* fcut, const float* dir, const float* Wf, const float* bf, int64_t N, int F,
                                 int R, float* dxc, float* dmu_in, float* dWf, float* dbf, float* workspace,
                                 int accumulate, hipStream_t stream);
/* The forward pass with the filter on the matrix pipe (F = 128, n_rbf in {8, 16, 20}): W = phi' Wf'^T as one small GEMM
 * per tile of 32 edge rows (the bias as a column of the contraction, the cutoff applied to the product as in the
 * reference, Geom3D/models/painn.py:54-58), the message arithmetic (:59-64) on the vector unit, a team
 * of four waves per molecule.  Rows are laid out in groups of four that share their target atom (the edges of an atom
 * in incidence order, padded to a multiple of four): row_edge [4 G] int32 (-1 = padding), grp_atom [G] int32 =
 * 2 * atom + (the group is the last one of its atom), mol_grp [B + 1] int32 = first group of a molecule; an atom
 * without edges has one group of padding rows.  Same sums as geossl_painn_interaction_fwd in another (fixed) order.
 * mu == NULL (round 6): mu is identically zero - the FIRST interaction (painn.py:249) - no row of it is staged or gathered;
 * likewise geossl_painn_interaction_bwd_mol[_skip] with mu == NULL and dmu_in == NULL (both or neither): the dmumu third of
 * the filter is not evaluated, its outputs are the exact zeros the general form computes, dmu_in is not written. */
int geossl_painn_interaction_fwd_mma(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                     const int32_t* row_edge, const int32_t* grp_atom, const int32_t* mol_grp,
                                     const float* phi, const float* fcut, const float* dir, const float* Wf,
                                     const float* bf, const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F, int R,
                                     float* q_out, float* mu_out, hipStream_t stream);
/* The same two passes with the molecule layout (mol_ptr [B+1] int32, max_n atoms in the largest molecule): one block
 * per molecule, the rows every edge of the molecule reads staged in LDS once (results identical bit for bit; falls back
 * to the per-atom kernels when F is not 64 / 128 or a molecule's rows do not fit the LDS).                      */
int geossl_painn_interaction_fwd_mol(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                     const int64_t* inc_ptr, const int32_t* inc_idx, const float* phi, const float* fcut,
                                     const float* dir, const float* Wf, const float* bf, const int32_t* mol_ptr,
                                     int64_t B, int max_n, int64_t N, int F, int R, float* q_out, float* mu_out,
                                     hipStream_t stream);
int64_t geossl_painn_interaction_bwd_mol_workspace_floats(int64_t N, int64_t B, int F, int R);
int geossl_painn_interaction_bwd_mol(const float* dq_out, const float* dmu_out, const float* mu, const float* xc,
                                     const int64_t* idx_i, const int64_t* inc_ptr, const int32_t* inc_idx,
                                     const float* phi, const float* fcut, const float* dir, const float* Wf,
                                     const float* bf, const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F,
                                     int R, float* dxc, float* dmu_in, float* dWf, float* dbf, float* workspace,
                                     int accumulate, hipStream_t stream);
int geossl_painn_mix_pre_fwd(const float* q, const float* mm, int64_t N, int F, float eps, float* ctx, float* dot,
                             hipStream_t stream);
/* mu_out == NULL (mix_post_fwd) / dmu_new == NULL (mix_post_bwd): the LAST block of the backbone - its mu' is nobody's
 * input (the representation is q, painn.py:262-269), so it is not formed and its gradient is the zero it is. */
int geossl_painn_mix_post_fwd(const float* q, const float* mu, const float* mm, const float* xx, const float* dot,
                              int64_t N, int F, float* q_out, float* mu_out, hipStream_t stream);
int geossl_painn_mix_post_bwd(const float* dq_new, const float* dmu_new, const float* mm, const float* xx,
                              const float* dot, int64_t N, int F, float* dxx, float* dmm, hipStream_t stream);
int geossl_painn_mix_pre_bwd(const float* dq_new, const float* dctx, const float* ctx, const float* mm, int64_t N, int F,
                             float* dq_in, float* dmm, hipStream_t stream);
int geossl_add(const float* a, const float* b, int64_t n, float* out, hipStream_t stream);

/* ---- PaiNN position gradient, first order — examples/finetune_md17.py:46 (grad(energy, positions)) through
 * Geom3D/models/painn.py:232-241,54-64.  edge_grads (one call per interaction block, in the backward's order, the first
 * with accumulate = 0): dphi [E][R], dfcut [E], ddir [E][3] += the block's dL/d(phi, fcut, dir) given the gradient at
 * the block's output (dq_out [N][F], dmu_out [N][3][F]), its inputs mu [N][3][F], xc [N][3F] and its filter_net rows
 * Wf [3F][R] / bf [3F].  edge_geom_bwd: dr [E][3] = dL/d r_ij (r_ij = pos[idx_i] - pos[idx_j]).  position_grad:
 * dpos [N][3] = sum of dr over the edges by idx_i minus the sum over the edges by idx_j (the two incidence lists of
 * geossl_incidence_*, sides = 1 and 2), fixed order. */
int geossl_painn_edge_grads(const float* dq_out, const float* dmu_out, const float* mu, const float* xc,
                            const int64_t* idx_i, const int64_t* idx_j, const float* phi, const float* fcut,
                            const float* dir, const float* Wf, const float* bf, int64_t E, int F, int R, float* dphi,
                            float* dfcut, float* ddir, int accumulate, hipStream_t stream);
int geossl_painn_edge_geom_bwd(const float* pos, const int64_t* idx_i, const int64_t* idx_j, int64_t E, float cutoff,
                               const float* offsets, const float* widths, int R, const float* dphi, const float* dfcut,
                               const float* ddir, float* dr, hipStream_t stream);
int geossl_painn_position_grad(const float* dr, const int64_t* inc_i_ptr, const int32_t* inc_i_idx,
                               const int64_t* inc_j_ptr, const int32_t* inc_j_idx, int64_t N, float* dpos,
                               hipStream_t stream);

/* ---- Adam — torch.optim.Adam step at pretrain_GeoSSL.py:258-260,343 over one flat fp32 buffer
 * (amsgrad off; weight_decay added to the gradient as torch does).  step_count is the 1-based step.  The
 * hyperparameters are the DOUBLES Python holds: torch forms 1 - beta, lr / bias_correction1 and sqrt(bias_correction2) in
 * double and rounds once; the update is torch's default (foreach) device arithmetic bit for bit
 * (tools/probes/adam_probe.hip).                                                                              */
int geossl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int64_t step_count, float grad_scale,
                     hipStream_t stream);

/* ---- Capacity launches: the `_dyn` entry points --------------------------------------------------------------------
 * The reference's loader is DataLoaderAtomTuple(dataset, batch_size, shuffle=True) over ragged molecules
 * (examples/pretrain_GeoSSL.py:301, Geom3D/dataloaders/dataloaders_AtomTuple.py:81-88): the atom, pair-slot and
 * super-edge counts of a batch change every step, while a captured HIP graph fixes every launch's grid and by-value
 * arguments.  A `_dyn` entry point is its namesake with the row count(s) ALSO readable from device memory: the by-value
 * count is the CAPACITY (it sizes the grid, the workspaces and the layer stride of [L][P][F] tensors), `dyn_*` (nullable;
 * NULL = the namesake's behaviour) points at an int32 holding the batch's real count, written before the launch reaches
 * the stream.  Rows at and past the real count do not exist: never read, never written; their blocks leave zero partial
 * sums.  Real counts must be >= 1.  The index structures a launch walks (mol_ptr, pair_ptr, work lists, incidence lists,
 * super_edge_index) are device data anyway.  geossl_amd.pretrain_GeoSSL.StepGraphs replays ONE graph per
 * (molecules per batch, capacity) on batches of any size sequence this way.
 *
 * dyn_view (the two NCSN entry points): both heads were handed the base address of ONE [view 0 ; view 1] feature /
 * gradient tensor; head 1's rows start *dyn_view rows (the real atom count of a view) behind head 0's.            */
#define GEOSSL_COPY_MAX 8
typedef struct GeosslCopyBatch {
  void* dst[GEOSSL_COPY_MAX];
  const void* src[GEOSSL_COPY_MAX];
  int64_t bytes[GEOSSL_COPY_MAX]; /* multiples of 4; buffers 4-byte aligned; src[i] == NULL: dst[i] is filled with zeros */
} GeosslCopyBatch;
/* batch.to(device) into the static inputs of a replayed graph (:248): n <= GEOSSL_COPY_MAX device copies, one launch */
int geossl_copy_n(const GeosslCopyBatch* batch, int n, hipStream_t stream);
/* perturb + both views + both super-edge length sets (:68-74,199-205); view 1 starts at row *dyn_N of pos2 / z2 */
int geossl_ddm_views_dyn(const float* pos, const float* noise, const int64_t* sei0, const int64_t* sei1, int64_t N,
                         int64_t S, float* pos2, float* d01, float* d02, const int64_t* z, int64_t z_stride, int64_t* z2,
                         const int32_t* dyn_N, const int32_t* dyn_S, hipStream_t stream);
int geossl_embedding_fwd_dyn(const int64_t* z, int64_t z_stride, const float* table, int num_classes, int64_t N, int F,
                             float* out, int32_t* status, const int32_t* dyn_N, hipStream_t stream);
int geossl_embedding_bwd_dyn(const int64_t* z, int64_t z_stride, const float* dh, int num_classes, int64_t N, int F,
                             float* dtable, float* workspace, int accumulate, const int32_t* dyn_N, hipStream_t stream);
int geossl_cfconv_filter_fwd_dyn(const float* pair_d, const float* pair_c, int64_t P, const GeosslFilterWeights* w, int L,
                                 int F, int G, const float* offset, float coeff, float* T, float* Wf,
                                 const int32_t* dyn_P, hipStream_t stream);
int geossl_cfconv_filter_bwd_dyn(const float* pair_d, const float* pair_c, const uint8_t* pair_flag,
                                 const int32_t* pair_i, const int32_t* pair_j, int64_t P, int64_t N,
                                 const GeosslFilterWeights* w, const GeosslFilterGradIn* g, int L, int F, int G,
                                 const float* offset, float coeff, const float* T, const GeosslFilterGradOut* out,
                                 float* workspace, int accumulate, const int32_t* dyn_P, const int32_t* dyn_N,
                                 hipStream_t stream);
/* work items [0, *dyn_nwork) of the list */
int geossl_cfconv_aggregate_work_dyn(const float* x, const float* Wf, const uint8_t* pair_flag, const int32_t* mol_ptr,
                                     const int32_t* pair_ptr, const int32_t* work, int64_t nwork, int max_n, int F,
                                     int swap, float* out, const int32_t* dyn_nwork, hipStream_t stream);
/* F = 128 (the weight-stationary chain kernel) only */
int geossl_linear_chain_dyn(const float* X, int ldx, const GeosslChain* chain, int64_t R, int F, const int32_t* dyn_R,
                            hipStream_t stream);
int geossl_linear_wgrad_dyn(const GeosslTnBatch* batch, int nprob, int64_t R, int M, int N, int lda, int ldb, int ldw,
                            float* workspace, int accumulate, const int32_t* dyn_R, hipStream_t stream);
int geossl_ddm_loss_fwd2_dyn(const GeosslNcsnHeadFwd* heads, const int64_t* batch, const int64_t* sei0,
                             const int64_t* sei1, int64_t S, int F, const int32_t* dyn_S, const int32_t* dyn_view,
                             hipStream_t stream);
int geossl_ddm_loss_bwd_fused2_dyn(const GeosslNcsnHeadBwd* heads, const int64_t* sei0, const int64_t* sei1, int64_t S,
                                   int64_t N, int F, const int64_t* stats_divisor, const float* gout,
                                   const int64_t* inc_ptr, const int32_t* inc_idx, int accumulate, const int32_t* dyn_S,
                                   const int32_t* dyn_view, hipStream_t stream);

/* ---- PaiNN on a capacity bucket (round 5): the precomputed radius_edge_index of a batch is geometry-dependent
 * (datasets_3D_Radius.py:120), so a replayed PaiNN step reads every edge structure from static buffers that ONE launch
 * rewrites per step.
 * geossl_painn_edge_layout: from the collated one-view edge list (rows src_i = radius_edge_index[0], src_j = [1], E edges
 * grouped by molecule in batch order, both ends in one molecule; dataloaders_AtomTuple.py:64-65) and the one-view
 * molecule CSR (mol_ptr [B + 1], N atoms) to the structures of the fused (clean | perturbed) 2B-molecule batch - the
 * perturbed view keeps the clean view's graph (pretrain_GeoSSL.py:190-191), its atoms start at N, its edges at E:
 *   idx_i2 / idx_j2 [2E]; incidence lists by idx_i (iptr_i [N2cap + 1], ilist_i [2E]) and by idx_j, an atom's edges in
 *   ascending edge order; atoms [2N, N2cap] get empty lists; the four-row group layout of the matrix-pipe forward
 *   (row_edge [4 G], grp_atom [G], G = geossl_painn_group_capacity(2 E_cap, N2cap)) with the groups of molecule m in
 *   [mol_grp[m], mol_grp_end[m]); *status is set to 1 on an edge that leaves its molecule (or a molecule above 256
 *   atoms) - such edges are left out.  Molecules of at most 256 atoms.
 * geossl_painn_interaction_fwd_mma_dyn: the namesake with mol_grp_end (NULL: groups lie back to back).
 * The element-wise launches take the real row / edge count from device memory like the other `_dyn` entry points.   */
int64_t geossl_painn_group_capacity(int64_t E2, int64_t N2);
int geossl_painn_edge_layout(const int64_t* src_i, const int64_t* src_j, int64_t E, const int32_t* mol_ptr, int64_t N,
                             int64_t B, int64_t N2cap, int64_t* idx_i2, int64_t* idx_j2, int64_t* iptr_i,
                             int32_t* ilist_i, int64_t* iptr_j, int32_t* ilist_j, int32_t* row_edge, int32_t* grp_atom,
                             int32_t* mol_grp, int32_t* mol_grp_end, int32_t* status, hipStream_t stream);
int geossl_painn_edge_geom_dyn(const float* pos, const int64_t* idx_i, const int64_t* idx_j, int64_t E, float cutoff,
                               const float* offsets, const float* widths, int R, float* dir, float* fcut, float* phi,
                               const int32_t* dyn_E, hipStream_t stream);
int geossl_painn_interaction_fwd_mma_dyn(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                         const int32_t* row_edge, const int32_t* grp_atom, const int32_t* mol_grp,
                                         const float* phi, const float* fcut, const float* dir, const float* Wf,
                                         const float* bf, const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F,
                                         int R, float* q_out, float* mu_out, const int32_t* mol_grp_end,
                                         hipStream_t stream);
/* Molecules above the LDS rows of a molecule-staged interaction launch (geossl_painn_stage_cap: kind 0 = matrix-pipe
 * forward, 1 = vector forward, 2 = backward; atoms) - Molecule3D with hydrogens has them - no longer send the whole
 * batch to the per-atom kernels: geossl_painn_interaction_fwd_mma_dyn and geossl_painn_interaction_bwd_mol_skip SKIP
 * such molecules, and the per-atom kernels cover a LIST of atoms (min(nlist, *dyn_nlist) entries of atom_list; the
 * backward with accumulate = 1 adds its filter-gradient partials to what the molecule-staged launch left in dWf / dbf;
 * workspace: geossl_painn_interaction_bwd_workspace_floats(nlist, F, R)).                                              */
int geossl_painn_stage_cap(int kind, int F, int R);
int geossl_painn_interaction_fwd_atoms(const float* q, const float* mu, const float* xc, const int64_t* idx_j,
                                       const int64_t* inc_ptr, const int32_t* inc_idx, const float* phi,
                                       const float* fcut, const float* dir, const float* Wf, const float* bf,
                                       const int32_t* atom_list, int64_t nlist, const int32_t* dyn_nlist, int F, int R,
                                       float* q_out, float* mu_out, hipStream_t stream);
int geossl_painn_interaction_bwd_atoms(const float* dq_out, const float* dmu_out, const float* mu, const float* xc,
                                       const int64_t* idx_i, const int64_t* inc_ptr, const int32_t* inc_idx,
                                       const float* phi, const float* fcut, const float* dir, const float* Wf,
                                       const float* bf, const int32_t* atom_list, int64_t nlist,
                                       const int32_t* dyn_nlist, int F, int R, float* dxc, float* dmu_in, float* dWf,
                                       float* dbf, float* workspace, int accumulate, hipStream_t stream);
int geossl_painn_interaction_bwd_mol_skip(const float* dq_out, const float* dmu_out, const float* mu, const float* xc,
                                          const int64_t* idx_i, const int64_t* inc_ptr, const int32_t* inc_idx,
                                          const float* phi, const float* fcut, const float* dir, const float* Wf,
                                          const float* bf, const int32_t* mol_ptr, int64_t B, int max_n, int64_t N, int F,
                                          int R, float* dxc, float* dmu_in, float* dWf, float* dbf, float* workspace,
                                          int accumulate, hipStream_t stream);
int geossl_painn_mix_pre_fwd_dyn(const float* q, const float* mm, int64_t N, int F, float eps, float* ctx, float* dot,
                                 const int32_t* dyn_N, hipStream_t stream);
int geossl_painn_mix_post_fwd_dyn(const float* q, const float* mu, const float* mm, const float* xx, const float* dot,
                                  int64_t N, int F, float* q_out, float* mu_out, const int32_t* dyn_N,
                                  hipStream_t stream);
int geossl_painn_mix_post_bwd_dyn(const float* dq_new, const float* dmu_new, const float* mm, const float* xx,
                                  const float* dot, int64_t N, int F, float* dxx, float* dmm, const int32_t* dyn_N,
                                  hipStream_t stream);
int geossl_painn_mix_pre_bwd_dyn(const float* dq_new, const float* dctx, const float* ctx, const float* mm, int64_t N,
                                 int F, float* dq_in, float* dmm, const int32_t* dyn_N, hipStream_t stream);

/* ---- Second-order route (training on forces): the primitives of geossl_amd/tape.py that are not dense products --------
 * Replaces, on the route that differentiates a force again (examples/finetune_md17.py:46-54: pred_force =
 * -grad(E, pos, create_graph=True); loss.backward()), the ATen element-wise / index kernels autograd would launch for
 * Geom3D/models/schnet.py:93,186,205-207,213-216 (distance, cosine envelope, Gaussian smearing, shifted softplus) and
 * Geom3D/models/painn.py:54-64,100-113, painn_utils.py:99-103,152-154 (gathers, index_add, split / cat, the xyz
 * broadcasts).  fp32, row-major [R][D]; PaiNN's [n][3][F] is the matrix [3 n][F].  No atomics: fixed summation order.
 * geossl_tape_unary:  y = f(alpha x + beta); kind: 0 affine, 1 exp, 2 cos, 3 sin, 4 shifted softplus, 5 sigmoid,
 *   6 s(1-s), 7 s(1-s)(1-2s), 8 reciprocal, 9 sqrt, 10 silu, 11 silu', 12 silu'', 13 exp(alpha x^2), 14 and 15 its first
 *   and second derivative in x, 16 (x < alpha), 17 |t|, 18 sign(t), 19 -1/t^2, 20 2/t^3,
 *   21 t^-1/2, 22 t^-3/2.
 * geossl_tape_binary: y[r][c] = scale * (A op B), op: 0 add, 1 sub, 2 mul, 3 A alone (a broadcast written out);
 *   operand modes: 0 [R][D], 1 [R] (per row), 2 [D] (per column), 3 [R/3][D] (the row r / 3; R a multiple of 3).
 * geossl_tape_reduce: the adjoints of those broadcasts: kind 1 row sums -> [R], 2 column sums -> [D] (workspace of
 *   geossl_tape_colsum_workspace_floats(R, D) floats), 3 sums over the xyz triple -> [R/3][D].
 * geossl_tape_gather_rows: out[e] = src[idx[e]] (idx int64 when idx64, else int32); geossl_tape_scatter_rows: its
 *   adjoint over a sorted incidence list: out[n] = sum of src[perm[k]], k in [ptr[n], ptr[n+1]) in ascending k (ptr
 *   int64 when ptr64, else int32 - the incidence lists of a PaiNN edge layout are taken as they are).
 * geossl_tape_copy2d: dst[r][c] = src[r][c] for an R x C block of two row-major matrices with row strides ld_*;
 * geossl_tape_fill: dst[i] = value.                                                                                  */
int geossl_tape_unary(int kind, const float* x, int64_t n, float alpha, float beta, float* y, hipStream_t stream);
int geossl_tape_binary(int op, const float* a, int amode, const float* b, int bmode, int64_t R, int D, float scale,
                       float* y, hipStream_t stream);
int64_t geossl_tape_colsum_workspace_floats(int64_t R, int D);
int geossl_tape_reduce(int kind, const float* x, int64_t R, int D, float* y, float* workspace, hipStream_t stream);
int geossl_tape_gather_rows(const float* src, const void* idx, int idx64, int64_t E, int D, float* out,
                            hipStream_t stream);
int geossl_tape_scatter_rows(const float* src, const void* ptr, int ptr64, const int32_t* perm, int64_t N, int D,
                             float* out, hipStream_t stream);
int geossl_tape_copy2d(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t R, int C,
                       hipStream_t stream);
int geossl_tape_fill(float* dst, int64_t n, float value, hipStream_t stream);

/* ---- batch assembly from a device-resident dataset (round 6) -------------------------------------------------------
 * Replaces, for molecules that already live in HBM, the per-step host collation of the reference's loader:
 * DataLoaderAtomTuple(dataset, batch_size, shuffle=True) (examples/pretrain_GeoSSL.py:295-301) ->
 * BatchAtomTuple.from_data_list (Geom3D/dataloaders/dataloaders_AtomTuple.py:46-73) over molecules that went through
 * AtomTupleExtractor (:15-37) and, for PaiNN, carry the radius_edge_index of datasets_3D_Radius.py:105-131.
 * One block per chosen molecule m (B of them, in batch order):
 *   x_dst / pos_dst rows [mol_ptr[m], mol_ptr[m+1]) = rows [src_off[m], ...) of x_src [*, x_cols] / pos_src [*, 3];
 *   batch_dst[row] = m (:61; NULL: skipped);
 *   sei0 / sei1 columns [se_ptr[m], se_ptr[m+1]) = the molecule's atom tuples + mol_ptr[m] (:64-65), option 0 =
 *     "combination" (itertools.combinations order), 1 = "permutation" (NULL: skipped);
 *   pair_i / pair_j = the pair-slot atoms of the TWO-VIEW batch (what geossl_pair_index_fill writes for the [2B+1]
 *     arrays [mol_ptr ; mol_ptr[1:] + N], pair_ptr2; N = mol_ptr[B]) (NULL: skipped);
 *   inc_idx entries [inc_ptr[a], inc_ptr[a+1]) of every atom a = its incident super-edges in ascending order (what
 *     geossl_incidence_fill(sides = 3) finds for the enumerations above) (NULL: skipped);
 *   e0_dst / e1_dst columns [e_ptr[m], e_ptr[m+1]) = columns [e_src_off[m], ...) of e0_src / e1_src with the node offset
 *     changed from src_off[m] to mol_ptr[m] (:64-65) (NULL: skipped).
 * zero (nullable): zero_count floats cleared by the same launch.  A collated batch is the special case src_off = mol_ptr.
 * Every pointer array is device memory written before the launch reaches the stream; nothing is read back.           */
typedef struct GeosslGather {
  const int64_t* x_src;
  const float* pos_src;
  const int32_t* src_off;   /* [B] */
  const int32_t* mol_ptr;   /* [B+1] */
  int64_t* x_dst;
  float* pos_dst;
  int64_t* batch_dst;
  const int32_t* se_ptr;    /* [B+1] */
  int64_t* sei0;
  int64_t* sei1;
  const int32_t* pair_ptr2; /* [2B+1] */
  int32_t* pair_i;
  int32_t* pair_j;
  const int64_t* inc_ptr;   /* [N+1] */
  int32_t* inc_idx;
  const int64_t* e0_src;
  const int64_t* e1_src;
  const int32_t* e_src_off; /* [B] */
  const int32_t* e_ptr;     /* [B+1] */
  int64_t* e0_dst;
  int64_t* e1_dst;
  float* zero;
  int64_t zero_count;
  int32_t x_cols;
  int32_t option;
} GeosslGather;
int geossl_gather_molecules(const GeosslGather* g, int64_t B, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOSSL_HIP_H */
