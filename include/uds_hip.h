/* uds_hip.h -- C ABI of libuds_hip.so, the MI355X (gfx950) message-passing engine for the
 * GNN-UDS graph-convolution hot path.
 *
 * The reference (Zhiyu014/GNN-UDS) has no FFI: its seam is the Keras layer protocol
 * (SURVEY.md section 8b).  Each entry point below replaces the TensorFlow ops behind one
 * reference call site (file:line relative to the reference's surrogate/ directory):
 *
 *   uds_dense_act            keras Dense                        emulator.py:198,203,206,212,225-226,278-279,313,317,324,329-330,336
 *                            + the "...NI,IHO->...NHO" node-update einsum of Spektral GATConv (via emulator.py:229-230)
 *   uds_csr_spmm             NodeEdge.call on its support       emulator.py:42-45 (used at :227-228,280-281)
 *                            + GCNConv propagation a_hat @ (xW) emulator.py:131-134
 *                            + post_proc_tf incidence matmuls   emulator.py:717-724
 *   uds_gat_forward          MixedGAT(GATConv)([x, adj])        emulator.py:18-25,229-230,282-283
 *   uds_spatial_layer_forward  one spatial-block loop body      emulator.py:225-230 / 278-283
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (fp32, row-major, contiguous,
 *     16-byte aligned); `rowptr`/`col` passed to uds_csr_create are HOST pointers and are copied;
 *   - snapshots: S = B*T independent graph signals share one graph; node features are (S, N, F);
 *   - all launches are asynchronous on `stream` (a hipStream_t; 0 = the null stream); the library
 *     never synchronises, never allocates after handle creation (graph-capture safe) and starts
 *     no host threads; handles are immutable after creation and may be shared across streams;
 *   - return value 0 = success, negative errno-style code otherwise; nothing throws across the
 *     boundary; uds_last_error() gives the thread-local message of the last failure.
 */
#ifndef UDS_HIP_H
#define UDS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UDS_ABI_VERSION 22

enum {
  UDS_OK = 0,
  UDS_EINVAL = -22, /* bad argument (shape, alignment, null pointer)   */
  UDS_ENOMEM = -12, /* device allocation failed                         */
  UDS_EHIP = -5,    /* a HIP runtime call failed (see uds_last_error)   */
  UDS_ENOSYS = -38  /* configuration not built into this library        */
};

/* keras.activations names used by the reference (emulator.py:66,193,324,330,336) */
enum {
  UDS_ACT_LINEAR = 0,
  UDS_ACT_RELU = 1,
  UDS_ACT_TANH = 2,
  UDS_ACT_SIGMOID = 3,
  UDS_ACT_HARD_SIGMOID = 4 /* Keras 2.10: clip(0.2*x + 0.5, 0, 1) */
};

typedef void *uds_stream_t; /* hipStream_t */

int uds_abi_version(void);
const char *uds_last_error(void);

/* ---- CSR pattern handle ------------------------------------------------------------------- */
typedef struct uds_csr uds_csr_t;

/* Copies a host CSR pattern (int32, rowptr[n_rows+1], col[nnz] < n_cols) to the device and builds
 * the degree-sorted row schedule (descending degree, ties by row index). */
int uds_csr_create(const int32_t *rowptr, const int32_t *col, int64_t n_rows, int64_t n_cols,
                   int64_t nnz, uds_csr_t **out);
int uds_csr_destroy(uds_csr_t *csr);
int uds_csr_shape(const uds_csr_t *csr, int64_t *n_rows, int64_t *n_cols, int64_t *nnz,
                  int32_t *max_degree);
/* Copies the row schedule (n_rows int32) to a HOST buffer: integer bookkeeping, tested bit-exact. */
int uds_csr_row_order(const uds_csr_t *csr, int32_t *out_host);

/* ---- single ops ---------------------------------------------------------------------------- */

/* out[r,:] = act([xa[r,:] | xb[r,:]] @ W + bias), r < rows.   W is (fa+fb, f_out) row-major.
 * xb may be NULL with fb = 0; bias may be NULL.  When a_self/a_nbr (f_out each) are given,
 * s_self[r] = <pre-activation row, a_self> and s_nbr[r] likewise are written too (the GAT
 * attention scalars); they must be NULL together. */
int uds_dense_act(const float *xa, int64_t fa, const float *xb, int64_t fb, int64_t rows,
                  const float *W, const float *bias, int64_t f_out, int act, const float *a_self,
                  const float *a_nbr, float *out, float *s_self, float *s_nbr,
                  uds_stream_t stream);

/* out[s,r,:] = act(sum_{p in row r} val[p] * x[s, col[p], :] + bias);  x is (S, n_cols, F),
 * out is (S, n_rows, F).  val (nnz) may be NULL (all ones); bias (F) may be NULL.  F % 4 == 0. */
int uds_csr_spmm(const uds_csr_t *csr, const float *val, const float *x, int64_t S, int64_t F,
                 const float *bias, int act, float *out, uds_stream_t stream);

/* keras Conv1D(H, taps, padding='causal', dilation_rate=dil, activation) along T on x laid out (B, T, R, F) with no
 * transposes: out[b,t,r,:] = act(sum_j x[b, t-(taps-1-j)*dil, r, :] @ kernel[j] + bias), zero outside [0, T).
 * dil < 0 looks ahead instead of back: with kernel[j] transposed this is the input gradient of the causal layer.
 * kernel is (taps, F, H) row-major (the Keras layout); out is (B, T, R, H).   emulator.py:155-157,244-257,299-310 */
int uds_conv1d_causal(const float *x, int64_t B, int64_t T, int64_t R, int64_t F, const float *kernel,
                      const float *bias, int64_t taps, int64_t dil, int64_t H, int act, float *out,
                      uds_stream_t stream);

/* The time recurrence of keras GRU / LSTM(H, return_sequences=True) (emulator.py:158-161, the `recurrent` alternatives to
 * the causal Conv1D), exact fp32.  xp (B, T, R, G*H) = x @ kernel + input bias for every time step (uds_dense_act), U
 * (H, G*H) the recurrent kernel, rb (G*H) the recurrent bias or NULL; kind 0 = GRU (G = 3, gate order z, r, h, TF2's
 * reset_after=True form, sigmoid recurrent activation), 1 = LSTM (G = 4: i, f, c, o).  out (B, T, R, H) = the hidden state
 * after every step, initial state zero.  H <= 256 and H * G*H floats must fit the LDS. */
int uds_recurrent_forward(const float *xp, const float *U, const float *rb, int64_t B, int64_t T, int64_t R, int64_t H,
                          int kind, float *out, uds_stream_t stream);

/* The same recurrence for a training step (emulator.py:457-484: fit_eval differentiates through the GRU / LSTM layers of
 * get_tem_nets, emulator.py:158-161; `recurrent: GRU` is the default of every block of utils/config.yaml): c_out (B, T, R, H)
 * additionally receives the LSTM's cell state after every step (NULL: not kept; ignored for the GRU).
 *
 * uds_recurrent_backward: back-propagation through time of a 64-unit layer, one launch on the matrix cores (split-bf16).
 * xp, h (the forward output), c (the LSTM's cell states, NULL for the GRU) and gh = dL/dh, all (B, T, R, .) as above;
 * packed = the G 64-column slices of U in uds_rowgemm_pack layout followed by the G slices of U_g^T (16 KiB each).
 * Outputs: dxp (B, T, R, G*64) = gradient of the input projection (the Dense backward takes it from there: dW = x^T dxp,
 * d b_in = sum dxp, dx = dxp W^T) and darec (G, B, T, R, 64) = gradient of the recurrent pre-activation h[t-1] U + b_rec,
 * gate-major so that dU[:, 64g:64g+64] = sum_t h[t-1]^T darec[g][t] is one uds_wgrad call with a time shift of 1 and
 * d b_rec its bias row. */
int uds_recurrent_forward_train(const float *xp, const float *U, const float *rb, int64_t B, int64_t T, int64_t R, int64_t H,
                                int kind, float *out, float *c_out, uds_stream_t stream);
int uds_recurrent_backward(const float *xp, const void *packed, const float *b_rec, const float *h, const float *c,
                           const float *gh, int64_t B, int64_t T, int64_t R, int kind, float *dxp, float *darec,
                           uds_stream_t stream);

/* A whole keras GRU / LSTM(64, return_sequences=True) layer in ONE launch on the matrix cores (split-bf16, three products,
 * fp32 accumulation): input projection, recurrent product, gates and state update per time step, the state fed back from
 * the accumulator registers; out (B, T, R, 64) = the hidden state after every step.   emulator.py:158-161 (`recurrent: GRU`
 * is the reference's default).
 *   F = 64 or 128: x (B, T, R, F) are the input rows; `packed` = uds_rowgemm_pack of the G 64-column slices of `kernel`
 *                  (F, G*64), then of `recurrent_kernel` (64, G*64), back to back; b_in (G*64) the input bias.
 *   F = 0:         x (B, T, R, G*64) is the input projection `x @ kernel + bias` itself (any input width, computed by
 *                  uds_rowgemm_forward_cat); `packed` holds the recurrent slices only; b_in is not read (for the GRU fold the
 *                  recurrent bias of the z and r gates into the projection's bias).
 * b_rec (G*64): the recurrent bias of the TF2 GRU (reset_after=True), NULL for the LSTM; kind 0 = GRU (G = 3: z, r, h),
 * 1 = LSTM (G = 4: i, f, c, o).  uds_recurrent_fused_supported(F, kind): whether W + U of that form fit the LDS. */
int uds_recurrent_fused_supported(int64_t F, int kind);
int uds_recurrent_fused(const float *x, int64_t F, const void *packed, const float *b_in, const float *b_rec, int64_t B,
                        int64_t T, int64_t R, int kind, float *out, uds_stream_t stream);

/* Matrix-core version of uds_dense_act (taps = 1, T = 1: rows = B*R) and uds_conv1d_causal for F % 32 == 0 and
 * f_out <= 64: operands split into bf16 hi + lo, three MFMA products, fp32 accumulation (the fused spatial kernel's
 * numerics).  `packed` = uds_rowgemm_pack(W (taps*F, f_out)) -- uds_rowgemm_packed_bytes() bytes, once per
 * parameter update.                                              emulator.py:155-157,313,317,324,329-330,336 */
int64_t uds_rowgemm_packed_bytes(int64_t k_total, int64_t f_out);
int uds_rowgemm_pack(const float *W, int64_t k_total, int64_t f_out, void *packed, uds_stream_t stream);
int uds_rowgemm_forward(const float *x, int64_t B, int64_t T, int64_t R, int64_t F, const void *packed,
                        const float *bias, int64_t taps, int64_t dil, int64_t f_out, int act, float *out,
                        uds_stream_t stream);

/* Two uds_rowgemm_forward problems of the same layer shape (B, T, F, taps, dil, f_out, act) in ONE launch when both are
 * small (<= 16 384 rows each: an autoregressive step on a 2k-node network), else as two launches: the node-side and the
 * link-side temporal layer of a step (emulator.py:244-257) -- problem i: x_i (B, T, R_i, F) -> out_i (B, T, R_i, f_out). */
int uds_rowgemm_forward_pair(const float *x0, int64_t R0, const void *packed0, const float *bias0, float *out0,
                             const float *x1, int64_t R1, const void *packed1, const float *bias1, float *out1, int64_t B,
                             int64_t T, int64_t F, int64_t taps, int64_t dil, int64_t f_out, int act, uds_stream_t stream);

/* uds_rowgemm_forward whose input row is the concatenation [x (F1) | x2 (F2)] of two tensors (x2 NULL, F2 = 0: one
 * tensor; two tensors need taps = 1) and whose f_out outputs go to columns [col0, col0 + f_out) of rows of `ldo`
 * floats: a 128-wide layer is two calls on the two column halves of its kernel, no concatenation copies
 * (`concat([x, NodeEdge(x_e)])` -> GATConv kernel, emulator.py:227-230 at embed_size = 128). */
int uds_rowgemm_forward_cat(const float *x, int64_t F1, const float *x2, int64_t F2, int64_t B, int64_t T, int64_t R,
                            const void *packed, const float *bias, int64_t taps, int64_t dil, int64_t f_out, int act,
                            float *out, int64_t ldo, int64_t col0, uds_stream_t stream);

/* spektral DiffusionConv as the reference runs it (dense mixed mode, emulator.py:135-138,229): channel q is
 * reduce_sum(polyval(theta_q, a_hat) @ x, -1) with tf.math.polyval applied to the ENTRIES of a_hat, so zero entries take the
 * constant coefficient c0[q] and the N x N product collapses to the CSR support:
 *   out[s, i, q] = act( c0[q] * tot[s] + sum_{p in row i} vals[p, q] * r[s, col[p]] ),
 * r[s, j] = sum_f x[s, j, f] (S, n_cols), tot[s] = sum_j r[s, j] (S), vals[p, q] = polyval(theta_q, a_p) - c0[q] (nnz, C),
 * out (S, n_rows, C), C % 4 == 0. */
int uds_diffusion_forward(const uds_csr_t *csr, const float *vals, const float *c0, const float *r, const float *tot,
                          int64_t S, int64_t C, int act, float *out, uds_stream_t stream);

/* Message buffers of the graph-sharded spatial block (one 200k-node network node-cut over the GPUs; the reference holds
 * whole graphs on one device, SURVEY.md F5 / 8e -- the exchange itself is torch.distributed isend / irecv over RCCL).
 *   pack:   buf[s, i, :] = i < nx ? x[s, idx_x[i], :] : e[s, idx_e[i - nx], :]     one buffer per peer, node rows then link rows
 *   unpack: the inverse scatter into rows idx_x of x and idx_e of e (the halo rows the peer owns).
 * x (S, n_x, F), e (S, n_e, F), buf (S, nx + ne, F), F % 4 == 0; idx_* are int32 device arrays of LOCAL row numbers. */
int uds_halo_pack(const float *x, int64_t n_x, const float *e, int64_t n_e, int64_t S, int64_t F, const int32_t *idx_x,
                  int64_t nx, const int32_t *idx_e, int64_t ne, float *buf, uds_stream_t stream);
int uds_halo_unpack(const float *buf, int64_t S, int64_t F, const int32_t *idx_x, int64_t nx, const int32_t *idx_e,
                    int64_t ne, float *x, int64_t n_x, float *e, int64_t n_e, uds_stream_t stream);

/* Dense remainder of a TRAINED NodeEdge layer.  The reference's layer is `(w * inci + b) @ x` with w, b dense trainable
 * (R, M) matrices (emulator.py:34-45); on the incidence support that is the CSR aggregation of the fused kernel, off the
 * support it is `rest @ x`, rest = b with the support entries zeroed -- a true (R x M) x (M x S*h) GEMM.  Matrix cores,
 * split-bf16 (three products, fp32 accumulation):
 *   out[s, r, :] = sum_m rest[r, m] * x[s, m, :]        x (S, M, h), out (S, R, h), h % 4 == 0, h <= 64.
 * `packed` = uds_remainder_pack(rest), uds_remainder_packed_bytes(R, M) bytes, once per parameter update; `workspace` =
 * uds_remainder_workspace_bytes(R, M, S, h) bytes of scratch per call (the transposed bf16 image of x, and the accumulator
 * pieces of the tiles k_remainder_gemm2 cuts along K: at most 128 MiB). */
int64_t uds_remainder_packed_bytes(int64_t R, int64_t M);
int uds_remainder_pack(const float *rest, int64_t R, int64_t M, void *packed, uds_stream_t stream);
int64_t uds_remainder_workspace_bytes(int64_t R, int64_t M, int64_t S, int64_t h);
int uds_remainder_forward(const void *packed, int64_t R, int64_t M, const float *x, int64_t S, int64_t h, void *workspace,
                          float *out, uds_stream_t stream);

/* The same with the activations computed on the way: out = rest @ act(e W + b) for e (S, M, F), F = 64 or 128, W (F x h, packed by
 * uds_rowgemm_pack), h = 32 or 64.  In a trained layer x_e = Dense(e) exists only to be multiplied by `rest` (emulator.py:225-228 with
 * :36-45): this entry writes it straight into the GEMM's bf16 operand planes instead of an fp32 tensor that is then read back, split
 * and transposed.  Same workspace as uds_remainder_forward. */
int uds_remainder_forward_dense(const void *packed, int64_t R, int64_t M, const float *e, int64_t F, const void *packed_w,
                                const float *bias, int act, int64_t S, int64_t h, void *workspace, float *out,
                                uds_stream_t stream);

/* keras Dense(64) (64 inputs) + prefix sum over time + residual + activation in one pass (matrix cores, split-bf16):
 *   out[b,t,r,:] = act( sum_{t' <= t} (x[b,t',r,:] @ kernel + bias) + res[b,0,r,:] ),   x, out: (B,T,R,64), res: (B,1,R,64) or NULL.
 * `packed` = uds_rowgemm_pack of the (64, 64) kernel.  The `dense_resx` layer and the `cumsum(x_out, axis=1) + res` that
 * follows it (emulator.py:313-320): one read and one write of the tensor instead of two each. */
int uds_dense_cumsum(const float *x, int64_t B, int64_t T, int64_t R, const void *packed, const float *bias,
                     const float *res, int act, float *out, uds_stream_t stream);

/* uds_dense_cumsum with the emulator's output heads as its epilogue (emulator.py:313-338): the 64-wide resnet output
 *   y = act(cumsum_t(x @ kernel + bias) + res)
 * is consumed where it is produced and never written: out[b,t,r,:] = [ act_a(y @ A + a_bias)  (n_a <= 4 columns)  |
 * act_f(h_n @ F + f_bias)  (1 column, only when n_hidden > 0) ] with h_0 = y, h_i = act_h(h_{i-1} @ H_i + h_bias_i), H_1
 * (64, 32), H_2 .. H_5 (32, 32; n_hidden <= 5): the `out` head + the flood chain on the node side (:324-333), the `e_out` head alone on the
 * link side (:336).  Every *_packed is uds_rowgemm_pack of that layer's kernel.  out (B, T, R, n_a + (n_hidden > 0)). */
typedef struct {
  const void *a_packed;    /* (64, n_a) */
  const float *a_bias;     /* n_a floats or NULL */
  const void *h_packed[5]; /* (64, 32), then up to four (32, 32); unused entries NULL */
  const float *h_bias[5];  /* 32 floats each or NULL */
  const void *f_packed;    /* (32, 1) */
  const float *f_bias;     /* 1 float or NULL */
  int32_t n_a, act_a, n_hidden, act_h, act_f;
} uds_heads_t;
int uds_dense_cumsum_heads(const float *x, int64_t B, int64_t T, int64_t R, const void *packed, const float *bias,
                           const float *res, int act, const uds_heads_t *heads, float *out, uds_stream_t stream);

/* out[b,t,r,:] = act(cumsum_t(x)[b,t,r,:] + res[b,0,r,:]); x, out (B,T,R,F), res (B,1,R,F) or NULL; F % 4 == 0.
 * The resnet head of the emulator.                                          emulator.py:313-320 */
int uds_cumsum_act(const float *x, const float *res, int64_t B, int64_t T, int64_t R, int64_t F, int act,
                   float *out, uds_stream_t stream);

/* spektral GlobalAttnSumPool in batch mode, the head of the RL agents' graph encoder (agent.py:93-94, `ConvNet`):
 * out[b, :] = sum_r softmax_r(<x[b, r, :], k>) x[b, r, :] for x (B, R, F), k (F), F a power of two up to 256.  One pass over the
 * rows (online softmax), fixed merge order: bitwise reproducible. */
int uds_attn_sum_pool(const float *x, const float *k, int64_t B, int64_t R, int64_t F, float *out, uds_stream_t stream);

/* keras Dropout in training mode (the emulator's Dropout(0.2) / Dropout(self.dropout) layers,   emulator.py:199-213,234-235,
 * 287-288,314-318; `self.model(inp, training=fit)`, :411,434): out[i] = x[i] / (1 - rate) if element i is kept, else 0, for n floats;
 * out may be x.  0 <= rate < 1.  The mask is a pure function of (seed, offset + i) -- Philox4x32-10, key = seed, counter =
 * (offset + i) / 4, word (offset + i) % 4, kept when word >= rate * 2^32 -- so the backward pass is the same call on the gradient
 * with the same (seed, offset) and nothing is stored; the caller advances offset by n per use. */
int uds_dropout(const float *x, int64_t n, float rate, uint64_t seed, uint64_t offset, float *out, uds_stream_t stream);

/* Link -> node flow balance of post_proc_tf: inc_n is the (N x E) incidence support, sign (nnz) its +1 / -1 values,
 * flow (S,E) the signed link flows; q_in, q_out (S,N) are scaled per node by scale_in / scale_out (N).
 * q_out = inc+ @ max(f,0) + inc- @ max(-f,0),  q_in = inc- @ max(f,0) + inc+ @ max(-f,0).   emulator.py:717-724 */
int uds_flow_balance(const uds_csr_t *inc_n, const float *sign, const float *flow, int64_t S,
                     const float *scale_in, const float *scale_out, float *q_in, float *q_out,
                     uds_stream_t stream);

/* One chunk of the autoregressive rollout after the forward (Emulator._model, emulator.py:403-423, with the edge-fusion
 * branch of post_proc_tf, :717-724): flow = ey[..., ce-1] * span_e + mini_e (de-normalised link flow, per link), q_in /
 * q_out as uds_flow_balance, preds (B,so,N,cy+2) = [y[...,0], q_in, q_out, y[...,1:]]; then both state windows are shifted
 * by `so` steps IN PLACE and fed with the prediction: x (B,T,N,cy+3) gets [preds with the last channel thresholded at 0.5
 * when flood != 0, b], ex (B,T,E,ce+1) gets [ey, 1].  y (B,so,N,cy), ey (B,so,E,ce), b (B,so,N,1).  Bit-identical to the
 * tensor-by-tensor composition. */
int uds_roll_update(const uds_csr_t *inc_n, const float *sign, const float *span_e, const float *mini_e,
                    const float *scale_in, const float *scale_out, const float *y, int64_t cy, const float *ey,
                    int64_t ce, const float *b, int64_t B, int64_t so, int64_t T, int flood, float *x, float *ex,
                    float *preds, uds_stream_t stream);

/* Floats of workspace uds_gat_forward needs: S * n * (d + 2). */
int64_t uds_gat_workspace_floats(int64_t n, int64_t S, int64_t d);

/* Single-head GATConv on a CSR pattern that already contains the self loops:
 *   hx = [xa | xb] @ W              (W: (fa+fb, d))
 *   alpha_ij = softmax_j leaky_relu_0.2( <hx_i, a_self> + <hx_j, a_nbr> ),  j in row i
 *   out_i = act( sum_j alpha_ij hx_j + bias )
 * xa:(S,n,fa), xb:(S,n,fb) or NULL, out:(S,n,d), d % 4 == 0. */
int uds_gat_forward(const uds_csr_t *graph, const float *xa, int64_t fa, const float *xb,
                    int64_t fb, int64_t S, const float *W, const float *a_self,
                    const float *a_nbr, const float *bias, int64_t d, int act, float *workspace,
                    float *out, uds_stream_t stream);

/* The attention / aggregation half of uds_gat_forward on its own: hx (S,n,d) = the transformed features, s_self / s_nbr
 * (S,n) = their projections on the two attention kernels, however the caller computed them (e.g. with the matrix-core
 * row GEMM for wide layers: d = 128 is the reference's default embed_size).
 *   out_i = act( sum_j softmax_j(leaky_relu_0.2(s_self_i + s_nbr_j)) hx_j + bias ),  j in row i of `graph`. */
int uds_gat_aggregate(const uds_csr_t *graph, const float *hx, const float *s_self, const float *s_nbr,
                      const float *bias, int64_t S, int64_t d, int act, float *out, uds_stream_t stream);

/* ---- reverse mode of the sparse operators (the GradientTape of fit_eval, emulator.py:457-484) ---------- */

/* uds_gat_aggregate with a per-snapshot edge mask: the `use_adj` variant, where the control action rewrites the adjacency
 * entries of the actuated links at every time step and GAT casts the result to int (a setting < 1 removes the entry) --
 * emulator.py:268-271,343-362.  edge_mask (S, nnz) floats in the pattern's entry order: entry p of snapshot s takes part
 * iff edge_mask[s, p] != 0 or it is the diagonal (spektral sets the diagonal to one after the rewrite). */
int uds_gat_aggregate_masked(const uds_csr_t *graph, const float *hx, const float *s_self, const float *s_nbr,
                             const float *bias, const float *edge_mask, int64_t S, int64_t d, int act, float *out,
                             uds_stream_t stream);

/* Attention part of uds_gat_forward, backwards.  With pre_i = sum_j alpha_ij hx_j (before bias / activation) and
 * grad = dL/dpre (S,n,d), hx / s_self / s_nbr as uds_gat_forward left them in its workspace:
 *   d_hx (S,n,d)   = dL/dhx (aggregation + both attention scores),
 *   ds_self, ds_nbr (S,n) = dL/ds_self, dL/ds_nbr  (for the gradients of the attention kernels: sum hx * ds),
 * graph_t = the transposed pattern as its own handle, perm_t (device, nnz int32) = for every entry of graph_t the
 * position of the same entry in graph's row-major order.  alpha_ws / de_ws: S * nnz floats of workspace each.
 * Replaces the tape through spektral GATConv._call_dense (emulator.py:229-230,282-283). */
int uds_gat_backward(const uds_csr_t *graph, const uds_csr_t *graph_t, const int32_t *perm_t, const float *grad,
                     const float *hx, const float *s_self, const float *s_nbr, const float *a_self,
                     const float *a_nbr, int64_t S, int64_t d, float *alpha_ws, float *de_ws, float *d_hx,
                     float *ds_self, float *ds_nbr, uds_stream_t stream);

/* The two entries above with Spektral's ATTENTION DROPOUT (GATConv: `attn_coef_drop = dropout(softmax(...))`, rate 0.5, which the
 * reference switches on together with the model's Dropout layers: `self.model(inp, training=fit)`, emulator.py:411,434):
 * coef (S, nnz) multiplies the normalised coefficient of entry p of snapshot s (0 or 1 / (1 - rate); uds_dropout on a tensor of
 * ones makes it).  Forward: out_i = act(sum_j alpha_ij coef_ij hx_j + bias); backward: the same workspace and outputs as
 * uds_gat_backward.  Training path only. */
int uds_gat_aggregate_coef(const uds_csr_t *g, const float *hx, const float *s_self, const float *s_nbr, const float *bias,
                           const float *coef, int64_t S, int64_t d, int act, float *out, uds_stream_t stream);
int uds_gat_backward_coef(const uds_csr_t *g, const uds_csr_t *gt, const int32_t *perm_t, const float *grad, const float *hx,
                          const float *s_self, const float *s_nbr, const float *a_self, const float *a_nbr, const float *coef,
                          int64_t S, int64_t d, float *alpha_ws, float *de_ws, float *d_hx, float *ds_self, float *ds_nbr,
                          uds_stream_t stream);

/* out[k] = sum_s <a[s, row(k), :], b[s, col(k), :]> for every entry k of the pattern (row-major order): the gradient
 * of the per-entry values of uds_csr_spmm (a = dL/dout (S,n_rows,F), b = x (S,n_cols,F)) -- NodeEdge.weight / bias
 * on the incidence support (emulator.py:34-45). */
int uds_csr_sddmm(const uds_csr_t *csr, const float *a, const float *b, int64_t S, int64_t F, float *out,
                  uds_stream_t stream);

/* Weight (and bias) gradient of a row GEMM / causal Conv1D tap on the matrix cores (split-bf16, 3 products):
 *   d_kernel[f, n] = sum_{b,t,r} a[b, t - shift, r, f] * g[b, t, r, n]   (terms with t < shift dropped),
 *   d_bias[n]      = sum_{b,t,r} g[b, t, r, n]                            (with_bias != 0),
 * a:(B,T,R,F), g:(B,T,R,H), F + (with_bias != 0) <= 128 and H <= 64 (uds_wgrad_workspace_floats() == 0 otherwise).
 * Dense layers: T = 1, shift = 0.  Deterministic (no atomics).  workspace: uds_wgrad_workspace_floats() floats.
 * Replaces the tape's `X^T dZ` products of emulator.py:457-484 (a library GEMM cannot split this reduction). */
int64_t uds_wgrad_workspace_floats(int64_t rows, int64_t F, int64_t H, int with_bias);
int uds_wgrad(const float *a, const float *g, int64_t B, int64_t T, int64_t R, int64_t F, int64_t H, int64_t shift,
              int with_bias, float *workspace, float *d_kernel, float *d_bias, uds_stream_t stream);

/* ---- one spatial layer (node side + link side) ---------------------------------------------- */
typedef struct uds_network uds_network_t;

/* Borrows the four patterns (they must outlive the network): adj (N x N) and edge_adj (E x E) with
 * self loops, inc_n (N x E) node<-links incidence support, inc_e (E x N) its transpose. */
int uds_network_create(const uds_csr_t *adj, const uds_csr_t *edge_adj, const uds_csr_t *inc_n,
                       const uds_csr_t *inc_e, uds_network_t **out);
int uds_network_destroy(uds_network_t *net);
/* uds_network_create plans the fused kernel for 64-wide node and link rows.  Layers with 96-wide inputs (the first
 * layer of block 2: H + d/2 features, emulator.py:260-262) need their own tile plans: build them here, once, before
 * the first forward with those widths (allocates; afterwards forwards stay allocation-free).  Without it such a layer
 * runs the unfused kernels.  fx / fe = input widths of the node / link side. */
int uds_network_prepare(uds_network_t *net, int64_t fx, int64_t fe);
/* info8 = {fused plan available, node tiles, link tiles, max primary rows per tile, max secondary rows per
 * tile, max metadata ints per tile, LDS bytes per workgroup, t_node*1000 + t_link}. */
int uds_network_plan_info(const uds_network_t *net, int32_t *info8);

/* Host-only access to the tile planner (integer bookkeeping; works without a GPU).  A plan clusters the
 * primary rows of each side (nodes under adj, links under edge_adj) into tiles of at most t_node / t_link
 * rows (clusters whose primary / secondary footprint exceeds p_limit / q_limit > 0 are bisected) and lists, per
 * tile, own rows, halo rows, secondary rows and local CSR indices (layout:
 * gnn_uds_amd/csrc/tile_plan.hpp).  hdr_out holds 8 ints per tile, pool_out pool_len ints. */
typedef struct uds_tile_plan uds_tile_plan_t;
int uds_tile_plan_create(const int32_t *adj_rowptr, const int32_t *adj_col, const int32_t *eadj_rowptr,
                         const int32_t *eadj_col, const int32_t *incn_rowptr, const int32_t *incn_col,
                         const int32_t *ince_rowptr, const int32_t *ince_col, int64_t n_node,
                         int64_t n_edge, int32_t t_node, int32_t t_link, int32_t p_limit,
                         int32_t q_limit, uds_tile_plan_t **out);
int uds_tile_plan_destroy(uds_tile_plan_t *plan);
int uds_tile_plan_sizes(const uds_tile_plan_t *plan, int64_t *n_tiles, int64_t *pool_len, int32_t *caps3);
int uds_tile_plan_copy(const uds_tile_plan_t *plan, int32_t *hdr_out, int32_t *pool_out);
/* The tile blocks the fused d = 64 kernel fetches (host-only, integer bookkeeping): one block per tile of the merged
 * list at a fixed stride of *stride_out ints, [8-int header | fixed-width index lists] (layout:
 * gnn_uds_amd/csrc/tile_plan.hpp, "Tile blocks of k_fused_tile").  Pass blocks_out = NULL to query the stride.  Returns
 * UDS_EINVAL when a tile exceeds the byte-wide local indices (255 primary / 256 secondary rows). */
int uds_tile_plan_blocks(const uds_tile_plan_t *plan, int32_t *blocks_out, int64_t *stride_out);

typedef struct uds_spatial_params {
  const float *xe_k, *xe_b; /* Dense(h) on e -> x_e : (fe, h), (h)            emulator.py:225 */
  const float *ex_k, *ex_b; /* Dense(h) on x -> e_x : (fx, h), (h)            emulator.py:226 */
  const float *ne_n_val;    /* NodeEdge(|inci|)   on its support: nnz(inc_n)  emulator.py:227 */
  const float *ne_e_val;    /* NodeEdge(|inci|^T) on its support: nnz(inc_e)  emulator.py:228 */
  const float *gx_k, *gx_as, *gx_an, *gx_b; /* GAT nodes: (fx+h, d),(d),(d),(d) emulator.py:229 */
  const float *ge_k, *ge_as, *ge_an, *ge_b; /* GAT links: (fe+h, d),(d),(d),(d) emulator.py:230 */
  const void *packed;       /* optional: output of uds_spatial_pack_weights for THESE kernels (device, 16-B aligned);
                               NULL = pack inside every forward call (4 tiny launches) */
} uds_spatial_params_t;

/* Bytes of the packed-weight buffer (fixed, covers fx, fe up to 96). */
int64_t uds_spatial_packed_bytes(void);
/* Splits the four kernels xe_k, ex_k, gx_k, ge_k into bf16 hi/lo MFMA fragments for the fused kernel.  Call once per
 * parameter update; the buffer must outlive the forward calls that reference it. */
int uds_spatial_pack_weights(const uds_spatial_params_t *params, int64_t fx, int64_t fe, int64_t h, int64_t d,
                             void *packed_out, uds_stream_t stream);

/* flags of uds_spatial_layer_forward */
enum {
  UDS_FLAG_EXACT_FP32 = 1,   /* exact-fp32 fmaf kernels (unfused); default is the fused kernel whose GEMMs
                                run as 3-term bf16-split MFMA with fp32 accumulation (~2^-16 per product) */
  UDS_FLAG_REQUIRE_FUSED = 2 /* fail instead of falling back when the fused kernel cannot take the shape */
};

/* Floats of workspace uds_spatial_layer_forward needs. */
int64_t uds_spatial_workspace_floats(const uds_network_t *net, int64_t S, int64_t h, int64_t d);

/* x:(S,N,fx), e:(S,E,fe) -> out_x:(S,N,d), out_e:(S,E,d).  out_* must not alias x / e.
 * Fused single-pass kernel for h = 32, d = 64, fx, fe in {64, 96}; other shapes (and UDS_FLAG_EXACT_FP32)
 * run the unfused exact-fp32 kernels. */
int uds_spatial_layer_forward(const uds_network_t *net, const uds_spatial_params_t *params,
                              const float *x, int64_t fx, const float *e, int64_t fe, int64_t S,
                              int64_t h, int64_t d, int act, int flags, float *workspace,
                              float *out_x, float *out_e, uds_stream_t stream);

/* The same layer with 96-wide inputs given as TWO tensors each: x = [xa (S,N,64) | xb (S,N,32)], e likewise (xb / eb
 * NULL with fxb / feb 0 = a single tensor).  The reference concatenates the boundary / action embeddings to the
 * temporal outputs before block 2 (emulator.py:260-262); here the concatenation is never materialised: the kernel
 * fetches a row's two pieces from the two tensors.  Fused kernel only (h = 32, d = 64). */
int uds_spatial_layer_forward_split(const uds_network_t *net, const uds_spatial_params_t *params, const float *xa,
                                    int64_t fxa, const float *xb, int64_t fxb, const float *ea, int64_t fea,
                                    const float *eb, int64_t feb, int64_t S, int64_t h, int64_t d, int act, int flags,
                                    float *workspace, float *out_x, float *out_e, uds_stream_t stream);

/* The same layer for a TRAINED NodeEdge: h = 64, d = 128, fx = 128, fe = 128 or 64 (the reference's default width) or h = 32,
 * d = 64, fx = fe = 64.  The reference's layer is matmul(w * inci + b, x) with a dense trainable b (emulator.py:36-45): after
 * training b is non-zero off the incidence support, and that part, rem_x = (b_n off the support) @ Dense_xe(e) (S,N,h) and
 * rem_e = (b_e off the support) @ Dense_ex(x) (S,E,h), is a dense GEMM (uds_remainder_forward).  Here it is added to the
 * support aggregate inside the fused kernel (one h-float row per primary row and snapshot, prefetched one snapshot ahead).
 * Fused kernel only: UDS_EINVAL when the shape or the network's tile plan does not take it. */
int uds_spatial_layer_forward_rem(const uds_network_t *net, const uds_spatial_params_t *params, const float *x, int64_t fx,
                                  const float *e, int64_t fe, const float *rem_x, const float *rem_e, int64_t S, int64_t h,
                                  int64_t d, int act, int flags, float *workspace, float *out_x, float *out_e,
                                  uds_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UDS_HIP_H */
