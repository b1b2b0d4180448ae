/* llark_hip.h -- C ABI of libllark_hip.so: the MI355X (gfx950) implementation of LLark's
 * audio-encoder -> LLM hot path.
 *
 * The reference (spotify-research/llark) has NO native/FFI boundary: its hot path is Python calling
 * third-party CUDA libraries.  This header is therefore the boundary a
 * maintainer of the reference would bind (via ctypes) underneath the
 * reference's own Python module interfaces.  Each entry point names the reference call site it
 * replaces.  Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - every function returns LLARK_OK (0) or a negative error code, never throws, and enqueues its
 *     work on `stream` (a hipStream_t passed as void*);
 *   - compute entry points never allocate and keep no hidden device state: the only functions that
 *     allocate are the two object constructors llark_workspace_create (1 KiB of device counters the
 *     persistent GEMM kernels synchronise through) and llark_vqvae_plan_create (a host-side launch
 *     list); both objects are owned, passed in and destroyed by the caller;
 *   - llark_last_error() returns a thread-local human-readable message for the last failure;
 *   - re-entrant per stream: a workspace serves ONE stream at a time (launches that share it are
 *     ordered on that stream); create one per (device, stream).  The remaining process-level state is
 *     immutable once set: per-kernel "dynamic LDS size raised" flags and occupancy figures of the code
 *     object, the CU count of each device ordinal.  Scratch memory a kernel needs beyond the workspace is
 *     passed in by the caller (llark_gemm16_fragw_sk);
 *   - the library reads NO environment variables: every choice it makes is a function of its arguments (tile variants,
 *     K-splitting, kernel forms); the only getenv in the sources is LLARK_LO8_PROF_BUF inside `#ifdef LLARK_LO8_PROF`,
 *     a cycle-counter build (scripts/build_lo8_prof.sh) that is never shipped.  The knobs of the Python host layer
 *     (LLARK_PRIOR_PRECISION, LLARK_LLM_PRECISION, LLARK_HIP_LIB, ...) are listed in INTEGRATION.md.
 */
#ifndef LLARK_HIP_H
#define LLARK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* llark_stream_t; /* hipStream_t */

enum {
    LLARK_OK = 0,
    LLARK_ERR_INVALID = -1,     /* bad argument (shape, null pointer, alignment) */
    LLARK_ERR_UNSUPPORTED = -2, /* shape outside what is compiled */
    LLARK_ERR_LAUNCH = -3       /* HIP launch / runtime error */
};

enum { LLARK_F16 = 0, LLARK_BF16 = 1 };

/* epilogues of llark_gemm16 */
enum {
    LLARK_EPI_F32 = 0,         /* C = acc + bias                              (fp32 out)        */
    LLARK_EPI_RESID = 1,       /* C = R + (acc + bias)                        (fp32, C may alias R) */
    LLARK_EPI_QGELU_SPLIT = 2, /* g = x*sigmoid(1.702x), x = acc+bias -> hi/lo 16-bit planes     */
    LLARK_EPI_OUT16 = 3,       /* out = 16-bit(acc + bias)                                       */
    LLARK_EPI_SWIGLU16 = 4,    /* out = 16-bit(silu(gate) * up), W rows interleaved [32 gate|32 up] */
    LLARK_EPI_SPLIT16 = 5,     /* acc + bias -> hi/lo 16-bit planes                              */
    LLARK_EPI_SWIGLU_SPLIT = 6, /* silu(gate)*up -> hi/lo 16-bit planes (fp32-class Llama mode)       */
    LLARK_EPI_QGELU_SPLIT8 = 7 /* llark_gemm16_lo8 only: g -> fp16 hi plane + e4m3 low plane          */
};

/* Caller-owned workspace of the persistent GEMM kernels (per-XCD chunk counters; see the conventions
 * above).  create() binds to the CURRENT device; returns NULL on failure (llark_last_error()). */
typedef struct llark_workspace* llark_workspace_t;
llark_workspace_t llark_workspace_create(void);
int llark_workspace_destroy(llark_workspace_t ws);

int llark_version(void);
const char* llark_last_error(void);
/* device properties probe: returns CU count (>0) or a negative error; arch_name receives e.g. "gfx950". */
int llark_device_info(int device, char* arch_name, int arch_name_len);

/* ---------------------------------------------------------------------------------------------
 * Audio front end: the sample-rate conversion inside `lr.load(fpath, sr=44100)` at jukebox/main.py:31 (librosa 0.7.2
 * -- the version openai/jukebox @ 08efbbc pins, docker/jukebox-embed.dockerfile:56 -- resamples with resampy's
 * "kaiser_best" band-limited sinc interpolation).  HOST entry point on HOST pointers, no stream, no device work:
 * x [n_in] -> y [n_out] at ratio = sr_new / sr_orig; win / dwin [nwin] = right half of the interpolation window with
 * num_table samples per zero crossing (scaled by ratio when ratio < 1) and its forward difference.
 * ------------------------------------------------------------------------------------------- */
int llark_resample_sinc_host(const float* x, int64_t n_in, double ratio, const double* win, const double* dwin, int nwin,
                             int num_table, float* y, int64_t n_out);

/* FLAC streams (the other container libsndfile -- `lr.load` at jukebox/main.py:31, `sf.read` at m2t/gcs_utils.py:123 -- reads without
 * further dependencies), decoded from memory on the HOST.  info: STREAMINFO fields (total_samples 0 = not recorded).  decode:
 * interleaved int32 [frames][channels] into out (NULL = count only), at most cap_frames frames; every frame's CRC-8 / CRC-16 is
 * checked and, with verify_md5, the MD5 signature of the decoded samples against STREAMINFO's (a FLAC stream certifies its decode). */
int llark_flac_info_host(const uint8_t* data, int64_t n, int* sample_rate, int* channels, int* bits_per_sample, int64_t* total_samples);
int llark_flac_decode_host(const uint8_t* data, int64_t n, int32_t* out, int64_t cap_frames, int64_t* decoded_frames, int verify_md5);

/* ---------------------------------------------------------------------------------------------
 * Jukebox VQ-VAE level-2 encoder: replaces `vqvae.encode(...)` at jukebox/main.py:61
 * (upstream openai/jukebox vqvae/encdec.py EncoderConvBlock, resnet.py ResConv1DBlock,
 * bottleneck.py BottleneckBlock.encode).  Activations: fp32 [n][C][T] (torch NCT).
 * ------------------------------------------------------------------------------------------- */
/* w[cout][cin][k] (torch Conv1d.weight) -> wp[k][cin][cout] (kernel layout). */
int llark_pack_conv_weight(const float* w, float* wp, int cout, int cin, int k, llark_stream_t stream);
/* nn.Conv1d forward; built shapes: (cin in {1,32,64}, cout 32, k4 s2) and (cin 32, cout 64, k3 s1). */
int llark_conv1d_f32(const float* x, int n, int cin, int tin, const float* wp, const float* bias, int cout, int k,
                     int stride, int pad, int dil, float* y, int tout, llark_stream_t stream);
/* ResConv1DBlock: y = x + conv1x1(relu(conv3(relu(x), dilation))); c must be 32. */
int llark_resblock_f32(const float* x, int n, int c, int t, const float* w1p, const float* b1, const float* w2p,
                       const float* b2, int dil, float* y, llark_stream_t stream);
/* kk[j] = sum_c k[j][c]^2 */
int llark_codebook_norms_f32(const float* k, int bins, int emb, float* kk, llark_stream_t stream);
/* BottleneckBlock.quantise: codes[n][t] = argmin_j |x[n][:,t] - k[j]|^2 (first minimum); min_dist optional. */
int llark_codebook_argmin(const float* x, int n, int emb, int t, const float* k, const float* kk, int bins,
                          int64_t* codes, float* min_dist, llark_stream_t stream);
/* The whole level-2 encode of jukebox/main.py:61 (`vqvae.encode(x)` -> zs[-1]) as ONE call: a plan lists the layers
 * (device pointers to llark_pack_conv_weight outputs and biases stay owned by the caller), llark_vqvae_encode runs
 * conv / residual blocks through two ping-pong buffers and finishes with the codebook search. */
void* llark_vqvae_plan_create(void);
void llark_vqvae_plan_destroy(void* plan);
int llark_vqvae_plan_add_conv(void* plan, const float* wp, const float* bias, int cin, int cout, int k, int stride, int pad);
int llark_vqvae_plan_add_resblock(void* plan, const float* w1p, const float* b1, const float* w2p, const float* b2, int width, int dil);
int llark_vqvae_encode(void* plan, const float* audio, int n, int t, float* buf0, float* buf1, long long buf_elems,
                       const float* codebook, const float* kk, int bins, int64_t* codes, int* t_out, llark_stream_t stream);
/* Fused stage of the level encoder on the 16-bit matrix cores (csrc/vqvae_fused.hip): one `down_t` step of upstream
 * EncoderConvBlock -- Conv1d(cin -> 32, k = 4, stride 2, pad 1), `depth` ResConv1DBlocks (dilations dil[0 .. depth), a HOST array),
 * optionally the block's output Conv1d(32 -> 64, k = 3, pad 1) -- in ONE launch, activations resident in LDS / registers, every
 * product as three fp16 MFMA passes over hi / lo splits of BOTH operands (fp32-class; the cin = 1 convolution stays an exact fp32
 * fmaf chain).  Replaces the same call site as llark_vqvae_encode (jukebox/main.py:61 -> VQVAE.encode); VQ codes equal the fp32
 * path's on every fixture, activations agree to ~1e-6 relative instead of bit for bit (the exact per-layer kernels stay).
 *   input : cin == 1: audio fp32 [n][tin];  cin == 32 / 64: planes in_hi / in_lo fp16 [n][tin][cin] (= a previous stage's output)
 *   weights: w0_f32 (cin == 1) = llark_pack_conv_weight layout [4][1][32]; every other weight operand = the hi / lo outputs of
 *            llark_vqvae_pack_frag16: w0 (32, cin, 4), wr = depth x [conv3 (32, 32, 3) | 1x1 (32, 32, 1) with perm1x1 = 1] concatenated
 *            (8 k-steps of 1 KiB per block and plane), wo (64, 32, 3) or NULL;  b0 [32], br [depth][2][32] (b1, b2), bo [64]
 *   output: planes out_hi / out_lo [n][tin/2][C] and / or out_f32 [n][C][tin/2], C = 64 with the output conv else 32.
 *   wexp  : HOST array of 2 + 2 depth ints: the exp2 each weight tensor was packed with (strided conv; dilated conv, 1x1 per block;
 *           output conv; 0 where a tensor is not fragment-packed).
 * tin even; sum(dil) (+1 with the output conv) <= 48. */
int llark_vqvae_stage_f16x2(const float* audio, const void* in_hi, const void* in_lo, int n, int cin, int tin, const float* w0_f32,
                            const void* w0_hi, const void* w0_lo, const float* b0, const void* wr_hi, const void* wr_lo,
                            const float* br, int depth, const int* dil, const void* wo_hi, const void* wo_lo, const float* bo,
                            const int* wexp, void* out_hi, void* out_lo, float* out_f32, llark_stream_t stream);
/* Conv1d.weight fp32 [cout][cin][k] (cout % 32 == 0, cin % 16 == 0) -> MFMA A fragments of w 2^exp2, fp16 hi / lo planes of
 * (cout / 32) * (k * cin / 16) * 512 elements each; perm1x1 != 0 (k == 1, cin == 32): channels in accumulator-register order.
 * exp2: choose max|w| 2^exp2 in [2^13, 2^14] so that the low plane is a normal fp16 (unscaled it is subnormal and the pair carries
 * ~19 bits instead of 22). */
int llark_vqvae_pack_frag16(const float* w, int cout, int cin, int k, int perm1x1, int exp2, void* hi, void* lo, llark_stream_t stream);

/* Near-tie certificate for the fused encoder (round 4; same call site: jukebox/main.py:61 -> VQVAE.encode -> bottleneck.encode).
 * llark_codebook_argmin plus: every token whose best / second-best distance gap is below  |x| (tie_a sqrt(d_best) + tie_b |x|)  is
 * appended to flag_list (ids n_index * t + token, at most `cap`; *flag_count -- zeroed by this call -- keeps counting past cap so the
 * caller can detect an overflow).  An encoder-output error e moves the gap by <= 2 |e| |k_a - k_b| <= 4 |e| sqrt(d): tie_a = 4 |e| / |x|;
 * the fp32 rounding of the distance chains moves it by a few ulp of |x|^2: tie_b = c * 2^-23. */
int llark_codebook_argmin_tie(const float* x, int n, int emb, int t, const float* k, const float* kk, int bins, int64_t* codes,
                              float tie_a, float tie_b, int* flag_count, int* flag_list, int cap, llark_stream_t stream);
/* Exact fix-up of the flagged tokens: for flag_list[0 .. count) (count read back by the caller) gathers a window of `win_tokens`
 * tokens of audio around each token (start clamp(tok - halo_tokens, 0, t_tok - win_tokens): a window edge on a clip edge is the
 * clip edge), runs the plan's exact per-layer kernels on the `count` windows and overwrites codes[] of the flagged tokens with the
 * argmin of the exact value -- bit-equal to llark_vqvae_encode on the whole clip when halo_tokens covers the receptive field.
 * win [count][win_tokens * raw_to_tokens] fp32, col [count] int, buf0 / buf1 (>= count * 32 * win_tokens * raw_to_tokens / 2 floats
 * each): caller-owned scratch.  count <= 0 is a no-op. */
int llark_vqvae_fix_near_ties(void* plan, const float* audio, int n, int t_samples, int raw_to_tokens, const int* flag_list, int count,
                              int halo_tokens, int win_tokens, float* win, int* col, float* buf0, float* buf1, long long buf_elems,
                              const float* codebook, const float* kk, int bins, int64_t* codes, llark_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Jukebox top prior, only_encode: replaces `top_prior.prior.forward(...)` at jukebox/main.py:108
 * and the pooling at jukebox/main.py:113-167.
 * ------------------------------------------------------------------------------------------- */
int llark_prior_embed(const int64_t* z, int n, int t, int width, int bins, const float* x_emb, const float* pos_emb,
                      const float* x_cond, const float* y_cond, float* h, llark_stream_t stream);
/* LayerNorm(width, eps) in fp32 -> fp16 hi/lo planes [rows][ldo] */
int llark_layernorm_split_f16(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta,
                              float eps, void* out_hi, void* out_lo, int ldo, llark_stream_t stream);
/* FactoredAttention core (pattern 1 block, 2 transpose-block, 3 previous-block) on qkv [n*t][ldq]. */
int llark_prior_attn(const float* qkv, int ldq, int n, int t, int n_state, int heads, int blocks, int pattern,
                     void* out_hi, void* out_lo, int ldo, llark_stream_t stream);
/* The same two producers in "lo8" mode (see llark_gemm16_lo8): fp16 hi plane [rows][ldo] + E4M3 low plane
 * [rows][ldo8] = fp8(sat((y - hi) * 2^sa)) in MFMA slot order; ldo8 in bytes, a multiple of 64 covering the width. */
int llark_layernorm_split_lo8(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta,
                              float eps, void* out_hi, int ldo, void* out_lo8, int ldo8, int sa, llark_stream_t stream);
int llark_prior_attn_lo8(const float* qkv, int ldq, int n, int t, int n_state, int heads, int blocks, int pattern,
                         void* out_hi, int ldo, void* out_lo8, int ldo8, int sa, llark_stream_t stream);
/* AvgPool1d(frame_len, stride=frame_len, ceil_mode=False) over time: h [n][t][width] -> out [n][frames][width] */
int llark_pool_window(const float* h, int n, int t, int width, int frame_len, float* out, int frames,
                      llark_stream_t stream);
/* acts[:len].mean(0): lens is a device int[n] (NULL -> t) */
int llark_pool_mean(const float* h, int n, int t, int width, const int* lens, float* out, llark_stream_t stream);
int llark_zero_pad16(void* plane, int rows, int ld, int from, llark_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 16-bit MFMA GEMM with fp32 accumulation: replaces upstream Conv1D.forward (`addmm`) of the prior
 * (split hi/lo fp16 activations, fp32-class accuracy) and nn.Linear of Llama / mm_projector
 * (m2t/models/llamav2.py:79,133,312).  C[m,n] = (a_hi [+ a_lo])[m,k] . wt[n,k]^T (+ bias).
 * ------------------------------------------------------------------------------------------- */
int llark_gemm16(int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda, const void* wt,
                 int ldw, const float* bias, int m, int n, int kp, float* c, int ldc, const float* resid, int ldr,
                 void* out_hi, void* out_lo, int ldo, llark_stream_t stream);
/* Same with an explicit tile variant (tuning knob: 0, 1, 2, 11, 12, 20 = persistent 12; -1 = library default). All
 * variants compute the same product; kp must be a multiple of 64 for the BK=64 variants (11, 12, 20). */
int llark_gemm16_ex(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                    const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc, const float* resid,
                    int ldr, void* out_hi, void* out_lo, int ldo, llark_stream_t stream);
/* llark_gemm16_ex plus a workspace: only with one are the persistent, chunk-synchronous variants (20 = 128x256x64,
 * 30 = the 256x256x64 LDS-DMA ring of csrc/gemm256.hip, 31 = the 256x256x64 tile with phases over N and resident A
 * fragments of csrc/gemm256n.hip -- the default for the prior's M = clips x 8192 split-fp16 products) chosen or honoured;
 * without, such requests run variant 12.  All variants produce bit-identical results. */
int llark_gemm16_ws(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                    const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc, const float* resid,
                    int ldr, void* out_hi, void* out_lo, int ldo, llark_workspace_t ws, llark_stream_t stream);

/* LayerNorm folded into the epilogues of the products around it (round 4; replaces the F.layer_norm + Conv1D pairs of upstream
 * transformer.py ResAttnBlock -- `attn(ln_0(x))`, `mlp(ln_1(x + a))` -- reached from jukebox/main.py:108).  One entry point, two roles:
 *   producer (ln_part != NULL, LLARK_EPI_RESID): c = resid + a.wt^T + bias as llark_gemm16_ws, plus out_hi / out_lo [m][ldo] = hi / lo of
 *            c * ln_vec[n] (ln_vec = gamma of the LayerNorm that follows) and ln_part [m][2 * ceil(n / 256)][2] = (sum, sum of squares) of c
 *            per 128-column slice; llark_ln_stats_finalize reduces them, in slice order, to ln_stat [m][2] = (mean, 1 / sqrt(var + eps));
 *   consumer (ln_stat != NULL, LLARK_EPI_F32 or LLARK_EPI_QGELU_SPLIT): a_hi / a_lo = those planes;
 *            out = rstd_m * (acc - mean_m * ln_vec[n]) + bias[n]  with ln_vec[n] = sum_k gamma_k wt[n][k], bias[n] = sum_k beta_k wt[n][k] + b[n]
 *            (both precomputed by the caller): == Conv1D(layer_norm(x)) without the LayerNorm kernel's read and write of x.
 * Split operands only; shapes the 256x256 tile does not take (llark_gemm16_ln_takes() == 0) return LLARK_ERR_UNSUPPORTED before anything
 * is launched.  Partial sums are written, never accumulated atomically: results are run-to-run bit-equal. */
int llark_gemm16_ln_takes(int m, int n, int kp);
int llark_gemm16_ln(int dtype, int epilogue, const void* a_hi, const void* a_lo, int lda, const void* wt, int ldw, const float* bias, int m,
                    int n, int kp, float* c, int ldc, const float* resid, int ldr, void* out_hi, void* out_lo, int ldo,
                    const float* ln_stat, const float* ln_vec, float* ln_part, llark_workspace_t ws, llark_stream_t stream);
int llark_ln_stats_finalize(const float* part, int rows, int nparts, int width, float eps, float* stat, llark_stream_t stream);
/* Round 5: the producer role with PREDICTED row statistics.  ln_pred [m][2] = (shift_m, scale_m), scale a power of two: out_hi / out_lo then
 * hold hi / lo of ((c - shift_m) scale_m) ln_vec[n] -- values of order one, so that rows whose level is far from their spread (|mean| >> std)
 * or whose spread is far from one (std ~ 1e-3: the fp16 lo plane would sit in its subnormals; std ~ 1e3: the hi plane would overflow) keep the
 * planes' 22 bits -- and ln_part the sums of (c - shift_m) and of its square.  llark_ln_stats_finalize_p reduces them to the pair the
 * UNCHANGED consumer role applies, ln_stat = ((mean - shift) scale, rstd / scale), and replaces ln_pred by the prediction for the next
 * LayerNorm of the same rows, (mean, nearest power of two of rstd).  llark_ln_row_pred makes the first prediction of a forward from the rows
 * of x themselves.  ln_pred = NULL: llark_gemm16_ln.  Same call sites (upstream ResAttnBlock ln_0 / ln_1 around the Conv1D products). */
int llark_gemm16_ln_p(int dtype, int epilogue, const void* a_hi, const void* a_lo, int lda, const void* wt, int ldw, const float* bias, int m,
                      int n, int kp, float* c, int ldc, const float* resid, int ldr, void* out_hi, void* out_lo, int ldo,
                      const float* ln_stat, const float* ln_vec, float* ln_part, const float* ln_pred, llark_workspace_t ws, llark_stream_t stream);
int llark_ln_stats_finalize_p(const float* part, int rows, int nparts, int width, float eps, float* stat, float* pred, llark_stream_t stream);
/* The producer role on 128x256 tiles, two workgroups to a CU, weights fragment-major (llark_pack_weight16_frag): one workgroup's epilogue
 * (residual loads, fp32 + plane stores) runs under the other's K loop, which the persistent 256x256 tile (one workgroup per CU) cannot do.
 * For the product with the short K loop -- upstream ResAttnBlock's attention-output Conv1D c_proj, K = n_state = 1200, reached from
 * jukebox/main.py:108 -- where that serial epilogue was 40 % of the launch.  Same arithmetic as llark_gemm16_ln_p's producer role (ln_pred may be
 * NULL); ln_part is [m][ceil(n / 64)][2] (64-column slices: pass nparts = ceil(n / 64) to llark_ln_stats_finalize[_p]).  kp % 64 == 0,
 * kp >= 192, n % 4 == 0; LLARK_ERR_UNSUPPORTED otherwise, before anything is launched. */
int llark_gemm16_lnp_fragw(int dtype, const void* a_hi, const void* a_lo, int lda, const void* wfrag, const float* bias, int m, int n, int kp,
                           float* c, int ldc, const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, const float* ln_vec,
                           float* ln_part, const float* ln_pred, llark_stream_t stream);
int llark_ln_row_pred(const float* x, int ldx, int rows, int width, float eps, float* pred, llark_stream_t stream);
/* "lo8" form of the prior's split GEMM (csrc/gemm256_lo8n.hip; OPT-IN reduced precision, the default is the two-pass fp16
 * form behind llark_gemm16_ws): same call site, upstream Conv1D.forward reached from
 * jukebox/main.py:108 with fp16=False.  The activation is a_hi = fp16(a) plus an E4M3 low plane
 * a_lo8[m][lda8] = fp8(sat((a - a_hi) * 2^sa)) whose 64-element k blocks are stored in MFMA slot order
 * (byte 32*(k/8 % 2) + 8*(k/16 % 4) + k % 8 of the block holds element k); the fp8 weight plane w8 = fp8(wt * 2^sw), same
 * slot order, is packed once by llark_pack_weight_lo8 (choose sw with max|W| * 2^sw <= 448; required).  C = a_hi.W^T + 2^-(sa+sw) a_lo8.W8^T (+ bias):
 * one v_mfma_scale_f32_32x32x64_f8f6f4 replaces the four fp16 MFMAs of the second pass.  Accuracy: the activation
 * carries 15-16 significant bits instead of 22 (measured end to end in tests/test_fulldepth_gpu.py: 36-layer embedding
 * max-abs-err 5.1e-4 against 5.4e-5 for the two-pass form -- above the 1e-4 of BASELINE configs[1], hence opt-in).
 * Operands are addressed with 32-bit byte offsets: m * lda * 2 and m * lda8 must stay below 2^31 (LLARK_ERR_UNSUPPORTED
 * otherwise; split the rows over several calls -- rows are independent).
 * epilogue: LLARK_EPI_F32, LLARK_EPI_RESID, LLARK_EPI_QGELU_SPLIT8 (out_hi fp16 [m][ldo] + out_lo8 [m][ldo8], same format).
 * kp % 64 == 0, kp >= 128; lda8 / ldo8 in bytes, multiples of 64. */
int llark_gemm16_lo8(int epilogue, const void* a_hi, const void* a_lo8, int lda, int lda8, const void* wt, int ldw,
                     const void* w8, int ldw8, const float* bias, int m, int n, int kp, int sa, int sw, float* c, int ldc, const float* resid, int ldr,
                     void* out_hi, void* out_lo8, int ldo, int ldo8, llark_workspace_t ws, llark_stream_t stream);
/* wt fp16 [n][ldw] -> out e4m3 [n][ldo] = fp8(sat(wt * 2^sw)) in MFMA slot order: the w8 operand of llark_gemm16_lo8. */
int llark_pack_weight_lo8(const void* wt, int ldw, int n, int kp, int sw, void* out, int ldo, llark_stream_t stream);
/* Fragment-major weights for the "B-direct" GEMM: wt [n][ldw] (16-bit, K-contiguous, kp % 64 == 0) -> dst of
 * ceil(n/32)*32 * kp elements laid out as 1-KiB chunks [row tile][k16 step][lane 0..63][8 elements] = one MFMA
 * B fragment per chunk (rows >= n are zero).  llark_gemm16_fragw computes the same product as llark_gemm16 but
 * streams these chunks L2 -> VGPR (the weight never goes through LDS); variant: -1 library choice,
 * 0 = 128x256 tiles, 1 = 128x128 tiles (no SwiGLU epilogues).   variant 2 (round 5, csrc/gemm_bda.hip; hi + lo bf16 operands, kp >= 192): the same 128x256 tiles
 * with A staged by LDS-DMA and the A fragments read one sub-step ahead -- results bit-identical to variant 0. */
int llark_pack_weight16_frag(const void* wt, int ldw, int n, int kp, void* dst, llark_stream_t stream);
int llark_gemm16_fragw(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                       const void* wfrag, const float* bias, int m, int n, int kp, float* c, int ldc, const float* resid,
                       int ldr, void* out_hi, void* out_lo, int ldo, llark_stream_t stream);
/* 1 when llark_gemm16_fragw_sk's library choice (variant -1, a scratch given) runs the product as whole 128x256 tiles of the per-tile
 * kernel -- no K cut, not the 128x128 tiles: the case in which llark_gemm16_fragw_rope_qkv (always whole 128x256 tiles) is BIT-equal to
 * llark_gemm16_fragw_sk + llark_rope_split_heads.  Shape rule only (no launch); callers ask it instead of restating the rule. */
int llark_gemm16_fragw_whole_tiles(int split, int epilogue, int m, int n, int kp);
/* The Llama q|k|v product of a PREFILL with RoPE, the head split, the K-cache append and the transposed V-cache append in its
 * epilogue: one launch instead of llark_gemm16_fragw (fp32 qkv) + llark_rope_split_heads (m2t/models/llamav2.py:224-234 -> HF
 * LlamaAttention q_proj / k_proj / v_proj, apply_rotary_pos_emb, cache update).  bf16 only; head_dim 128, nh even, s >= 32.
 * a_hi / a_lo [batch * s][lda]: RMSNorm output planes (a_lo NULL = plain bf16 mode: the three *_lo outputs must be NULL too).
 * wfrag: llark_pack_weight16_frag of the [3 * nh * 128][kp] q|k|v weight with the q and k rows of every head reordered to
 * [0..31 | 64..95 | 32..63 | 96..127] (v rows natural), so that a rotation pair (d, d + 64) shares a lane and register.
 * Outputs as llark_rope_split_heads writes them; bit-equal to the two-launch path wherever that runs whole 128x256 tiles. */
int llark_gemm16_fragw_rope_qkv(const void* a_hi, const void* a_lo, int lda, const void* wfrag, int kp, int batch, int s, int nh, int hd,
                                int pos0, const float* cos_t, const float* sin_t, int max_pos, void* q, void* k_cache, void* vt_cache,
                                void* q_lo, void* k_cache_lo, void* vt_cache_lo, int smax, llark_stream_t stream);
/* llark_gemm16_fragw with the K range of a tile cut over several workgroups, for products whose tile count is not a whole
 * number of rounds of the resident workgroups (Llama prefill at M = 2968: 384 / 1152 / 2064 tiles of 128x256 against 512
 * resident workgroups; one clip, M = 371: 48 .. 258 tiles -- m2t/models/llamav2.py:224-234 -> the q/k/v, o, gate/up, down
 * projections of HF LlamaDecoderLayer).  variant -1: where the measured rule says it pays (gemm.hip, gemm16_fragw_impl), the
 * library's choice cuts EVERY tile into the same 2 or 4 K ranges (one workgroup each; the last range finishes the tile and
 * adds the others' fp32 partial tiles from `scratch` in order: deterministic, no atomics on data, no assumption about how
 * many workgroups are resident); variant 0 instead cuts the ragged remainder of the tile order into equal runs over exactly
 * the resident workgroups (stream-K; needs an otherwise idle device).  Same product and epilogues as llark_gemm16_fragw (QGELU_SPLIT
 * excepted, which runs the per-tile kernel), equal up to the fp32 summation order over K.
 * scratch: caller-owned device memory, >= llark_gemm16_sk_scratch_bytes() bytes (the function returns that size for the
 * CURRENT device, -1 on error), 16-byte aligned, ZEROED once after allocation; launches sharing it must be ordered on one
 * stream.  The hand-off flags occupy the LAST 64 KiB of [scratch, scratch + scratch_bytes) whatever the shape of a launch (the
 * fp32 partial tiles grow from the front and are checked never to reach them), so pass the same (scratch, scratch_bytes)
 * pair on every call that shares a scratch.  scratch == NULL, or a problem too small to cut -> exactly llark_gemm16_fragw. */
long long llark_gemm16_sk_scratch_bytes(void);
int llark_gemm16_fragw_sk(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda,
                          const void* wfrag, const float* bias, int m, int n, int kp, float* c, int ldc, const float* resid,
                          int ldr, void* out_hi, void* out_lo, int ldo, void* scratch, long long scratch_bytes,
                          llark_stream_t stream);
/* Decode step (m <= 16 rows, bf16): h[m][n] += a . wt^T and then RMSNorm(h; norm_w, eps) -> bf16 planes x_hi (/x_lo),
 * in one launch (the last workgroup to finish normalises the complete rows; bit-identical to llark_gemm16 +
 * llark_rmsnorm_bf16).  Replaces o_proj / down_proj + the next LlamaRMSNorm of the cached decode path
 * (m2t/models/llamav2.py:224-234 -> HF LlamaDecoderLayer). */
int llark_gemm16_resid_rmsnorm(int dtype, int split, const void* a_hi, const void* a_lo, int lda, const void* wt, int ldw,
                               int m, int n, int kp, float* h, int ldh, const float* norm_w, float eps, void* x_hi,
                               void* x_lo, int ldx, llark_stream_t stream);
/* Decode step (m <= 16 rows, bf16): RMSNorm(x; norm_w, eps) . wt^T with the norm fused INTO the consuming weight-streaming
 * GEMM (every workgroup re-derives the row scales; bit-identical to llark_rmsnorm_bf16 + llark_gemm16).  epilogue: F32 or
 * SwiGLU(16/split).  Replaces LlamaRMSNorm + q/k/v, gate/up and lm_head projections of the cached decode path. */
int llark_gemm16_rmsnorm_a(int dtype, int split, int epilogue, const float* x, int ldx, const float* norm_w, float eps,
                           const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc, void* out_hi,
                           void* out_lo, int ldo, llark_stream_t stream);
/* Batched form (grid.y = batch): per-batch element strides for A, wt, c and the 16-bit outputs (no bias / residual).
 * Used by the attention backward of the training step (one product per (sequence, head)). */
int llark_gemm16_batched(int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda, long long stride_a,
                         const void* wt, int ldw, long long stride_w, int m, int n, int kp, float* c, int ldc,
                         long long stride_c, void* out_hi, void* out_lo, int ldo, long long stride_o, int batch,
                         llark_stream_t stream);
/* w[k][n] (upstream Conv1D.w, 16-bit) -> wt[n][ldw] with zero K padding; also serves row-major copies
 * (transpose=0: w is already [n][k], e.g. nn.Linear.weight). src_dtype/dst_dtype: LLARK_F16/BF16 or
 * 2 for fp32 source. */
int llark_pack_weight16(const void* w, int src_dtype, int transpose, int k, int n, void* wt, int dst_dtype, int ldw,
                        llark_stream_t stream);
/* fp32 [rows][ld_in] -> 16-bit hi/lo planes [rows][ldo] (lo may be NULL); pads [width, ldo) with zeros. */
int llark_split16(int dtype, const float* x, int ldx, int rows, int width, void* out_hi, void* out_lo, int ldo,
                  llark_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Llama-2 decoder: what WrappedLlamav2Model.forward delegates to HF LlamaModel.forward
 * (m2t/models/llamav2.py:224-234) plus the embed_tokens gather (m2t/models/llamav2.py:124).
 * Residual stream fp32 [rows][hidden]; Linear inputs / q / k / v / probabilities bf16.
 * ------------------------------------------------------------------------------------------- */
/* nn.Embedding gather: out[row] = table[ids[row]] (table_dtype LLARK_F16 / LLARK_BF16 / 2 = fp32). */
int llark_embed_gather(const int64_t* ids, int rows, const void* table, int table_dtype, int vocab, int width,
                       float* out, int ldo, llark_stream_t stream);
/* LlamaRMSNorm -> bf16 (out_lo optional second plane with the rounding residual). */
int llark_rmsnorm_bf16(const float* x, int ldx, int rows, int width, const float* w, float eps, void* out_hi,
                       void* out_lo, int ldo, llark_stream_t stream);
/* apply_rotary_pos_emb (half-split) + split heads + KV-cache write.  qkv fp32 [batch*s][3*nh*hd];
 * q bf16 [batch][nh][s][hd]; k_cache bf16 [batch][nh][smax][hd]; vt_cache bf16 [batch][nh][hd][smax]
 * (V transposed); rows/columns pos0..pos0+s-1 are written; cos/sin fp32 [max_pos][hd/2]. */
/* q_lo / k_cache_lo / vt_cache_lo: optional second planes holding the bf16 rounding residual (all three or
 * none): the "fp32-class" mode in which q, k, v keep 16 significant bits. */
int llark_rope_split_heads(const float* qkv, int batch, int s, int nh, int hd, int pos0, const float* cos_t,
                           const float* sin_t, int max_pos, void* q, void* k_cache, void* vt_cache, void* q_lo,
                           void* k_cache_lo, void* vt_cache_lo, int smax, llark_stream_t stream);
/* causal attention over the cache: query i sees keys j <= past + i. out bf16 [batch*s][nh*hd]. */
int llark_attn_prefill_bf16(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                            const void* k_cache_lo, const void* vt_cache_lo, int batch, int s, int nh, int hd, int past,
                            int smax, void* out, void* out_lo, llark_stream_t stream);
/* single-token attention over `total` cached keys. q bf16 [batch][nh][hd]; out bf16 [batch][nh*hd]. */
int llark_attn_decode_bf16(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                           const void* k_cache_lo, const void* vt_cache_lo, int batch, int nh, int hd, int total, int smax,
                           void* out, void* out_lo, llark_stream_t stream);
/* Same two kernels with ALiBi (MPT, m2t/llava/model/mpt/attention.py:build_alibi_bias + :58): alibi_slopes fp32 [nh]
 * (nullptr = none); slope_h * (key - (keys_visible - 1)) is added to the scaled scores before the softmax. */
int llark_attn_prefill_bf16_alibi(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                                  const void* k_cache_lo, const void* vt_cache_lo, int batch, int s, int nh, int hd, int past,
                                  int smax, void* out, void* out_lo, const float* alibi_slopes, llark_stream_t stream);
int llark_attn_decode_bf16_alibi(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                                 const void* k_cache_lo, const void* vt_cache_lo, int batch, int nh, int hd, int total, int smax,
                                 void* out, void* out_lo, const float* alibi_slopes, llark_stream_t stream);

/* Decode-step Linear: c[m][n] = sum_k a[m][k] wt[n][k] (+ bias[n]) for m <= 4 activation rows, bf16 operands (a = hi plane + optional
 * lo plane for the fp32-class mode), fp32 accumulate -- the weight-streaming form of llark_gemm16 (csrc/gemv_dma.hip: weights global ->
 * LDS by LDS-DMA, 96 KiB in flight per CU, no barrier in the K loop).  Replaces nn.Linear under LlamaModel.forward with one new token
 * (m2t/infer.py:146 -> model.generate; transformers==4.29.2 modeling_llama.py).  epilogue: LLARK_EPI_F32, LLARK_EPI_RESID (resid may
 * alias c), LLARK_EPI_SWIGLU16 / LLARK_EPI_SWIGLU_SPLIT (wt rows interleaved [gate 32 | up 32], out [m][n / 2]).  kp % 8 == 0,
 * kp <= 12288, lda / ldw % 8 == 0; other shapes return LLARK_ERR_UNSUPPORTED (use llark_gemm16). */
int llark_gemv16_dma(int split, int epilogue, const void* a_hi, const void* a_lo, int lda, const void* wt, int ldw, const float* bias,
                     int m, int n, int kp, float* c, int ldc, const float* resid, int ldr, void* out_hi, void* out_lo, int ldo,
                     llark_stream_t stream);
/* ... with LlamaRMSNorm fused in front (decode: input_layernorm -> q/k/v_proj, post_attention_layernorm -> gate/up_proj, norm ->
 * lm_head; modeling_llama.py LlamaRMSNorm): a = bf16 hi (+ lo when split) of norm_w * (x * rstd), x fp32 [m][ldx].  Bit-identical to
 * llark_rmsnorm_bf16 followed by llark_gemv16_dma.  m == 1, kp <= 4096; epilogue LLARK_EPI_F32 or SwiGLU. */
int llark_gemv16_dma_rmsnorm(int split, int epilogue, const float* x, int ldx, const float* norm_w, float eps, const void* wt, int ldw,
                             const float* bias, int m, int n, int kp, float* c, int ldc, void* out_hi, void* out_lo, int ldo,
                             llark_stream_t stream);

/* Backward products of nn.Linear without transposed copies of their operands (csrc/gemm_tn.hip; torch autograd of the Linear
 * layers under WrappedLlamav2ForCausalLM.forward + loss.backward(), m2t/models/llamav2.py:259-337, m2t/train.py:53-277):
 *   c[m][n] (= | +=) sum_k A(m, k) W(n, k), 16-bit operands, fp32 accumulate / output.
 * trans_a == 0: a is [m][lda], k contiguous; trans_a != 0: a is [kp][lda], row = k and column = m (e.g. dY for dW = dY^T X).
 * trans_b == 0: wt is [n][ldw], k contiguous; trans_b != 0: wt is [kp][ldw], row = k and column = n (e.g. W itself for dX = dY W).
 * kp % 64 == 0 and every one of the kp contraction rows / columns is read (pad with zeros); lda, ldw % 8 == 0; a transposed
 * operand's free size % 8 == 0.  epilogue: LLARK_EPI_F32 or LLARK_EPI_RESID (resid may alias c). */
int llark_gemm16_t(int dtype, int epilogue, int trans_a, int trans_b, const void* a, int lda, const void* wt, int ldw, int m, int n,
                   int kp, float* c, int ldc, const float* resid, int ldr, llark_stream_t stream);
/* ... and *sumsq (a double in device memory) += the sum of squares of every value written to c: the dW product of the last
 * micro-batch leaves its share of the squared gradient norm behind (HF Trainer's clip_grad_norm_; llark_adamw_clip consumes it). */
int llark_gemm16_t_sumsq(int dtype, int epilogue, int trans_a, int trans_b, const void* a, int lda, const void* wt, int ldw, int m, int n,
                         int kp, float* c, int ldc, const float* resid, int ldr, double* sumsq, llark_stream_t stream);

/* llark_gemm16_t / llark_gemm16_t_sumsq (sumsq nullable) with an explicit tile / pipeline variant, for benchmarks and A/B tests:
 * -1 = library choice, 0 = one LDS stage per workgroup, 1 = 128x256x32 tiles with two stages, 2 = 128x256x64 two stages, 3 = 256x256x64
 * two stages (8 waves).  Results are bit-identical across variants. */
int llark_gemm16_t_ex(int variant, int dtype, int epilogue, int trans_a, int trans_b, const void* a, int lda, const void* wt, int ldw, int m,
                      int n, int kp, float* c, int ldc, const float* resid, int ldr, double* sumsq, llark_stream_t stream);

/* Training pair (csrc/llama.hip, csrc/attn_bwd.hip): what torch autograd does for LlamaAttention's eager path
 * (transformers==4.29.2 modeling_llama.py) under WrappedLlamav2ForCausalLM.forward + loss.backward()
 * (m2t/models/llamav2.py:259-337, m2t/train.py:53-277), without ever writing an S x S matrix.
 * Forward: bf16 operands, no past keys; also writes lse fp32 [batch*nh][s], the log-sum-exp of each query's scaled masked scores. */
int llark_attn_prefill_bf16_lse(const void* q, const void* k_cache, const void* vt_cache, int batch, int s, int nh, int hd,
                                int smax, void* out, float* lse, const float* alibi_slopes, llark_stream_t stream);
/* Backward: q, dO, v_rm bf16 [batch*nh][s][128] (row-major); k_cache bf16 [batch*nh][smax][128]; o bf16 [batch*s][nh*128] = the
 * forward's out; lse from the forward; dsum fp32 [batch*nh][s] scratch.  Outputs dq, dk, dv fp32 [batch*nh][s][128] (before the RoPE
 * backward / head merge).  No transposed copies of any operand are needed (transposing LDS reads inside).
 * alibi_slopes (both calls): nullptr (Llama) or fp32 [nh] (MPT-1B trainer), the bias of llark_attn_prefill_bf16_alibi. */
int llark_attn_backward_bf16(const void* q, const void* k_cache, const void* v_rm, const void* dO, const void* o, const float* lse,
                             float* dsum, int batch, int s, int nh, int hd, int smax, float* dq, float* dk, float* dv,
                             const float* alibi_slopes, llark_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * MPT backbone (m2t/models/mpt.py over m2t/llava/model/mpt/{blocks,attention,norm}.py): the row-wise / element-wise
 * pieces that differ from Llama.  LayerNorm (norm.py; beta may be NULL for no_bias models) to bf16 hi (+ lo) planes
 * for the next GEMM, or fp32 -> fp32 (qk_ln over the q / k column blocks of the fused qkv, attention.py:330-333);
 * clip_qkv (attention.py:325-326); exact erf GELU between up_proj and down_proj (blocks.py:15,19) to bf16 planes.
 * ------------------------------------------------------------------------------------------- */
int llark_layernorm_bf16(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta, float eps,
                         void* out_hi, void* out_lo, int ldo, llark_stream_t stream);
int llark_layernorm_f32(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta, float eps,
                        float* y, int ldy, llark_stream_t stream);
int llark_clamp_f32(float* x, long long n, float limit, llark_stream_t stream);
/* backward of that clamp in the MPT training step: dy (bf16 [n], in place) keeps its value where |x| <= limit (x = the qkv
 * BEFORE the clamp) and becomes 0 elsewhere -- torch.clamp's gradient under loss.backward() (m2t/train.py:53-277). */
int llark_clamp_bwd_bf16(const float* x, long long n, float limit, void* dy, llark_stream_t stream);
int llark_scale_f32(float* x, long long n, float a, llark_stream_t stream);   /* logits *= logit_scale */
int llark_gelu_split_bf16(const float* x, int ldx, int rows, int width, void* out_hi, void* out_lo, int ldo,
                          llark_stream_t stream);
/* Backward pieces of the MPT training step: LayerNorm (dx [+= if accumulate], dgamma / dbeta accumulated with atomics),
 * exact GELU (fp32 copy optional, bf16 copy for the GEMMs), causal softmax rows with the ALiBi bias (probabilities of the
 * materialised attention backward; slopes fp32 [nh], row b of the batch belongs to head b % nh). */
int llark_layernorm_bwd(const float* x, int ldx, const float* gamma, const float* dy, int ldy, int rows, int width, float eps,
                        float* dx, int lddx, float* dgamma, float* dbeta, int accumulate, llark_stream_t stream);
int llark_gelu_bwd(const float* up, const float* dact, long long n, float* dup32, void* dup16, llark_stream_t stream);
int llark_causal_softmax_rows_alibi(const float* scores, int batch, int s, float scale, const float* slopes, int nh, void* p_out,
                                    int ldp, llark_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * CLAP HTSAT-base audio encoder (scripts/clap/clap_embeddings.py:63-107 -> laion_clap
 * CLAP_Module(enable_fusion=False, amodel="HTSAT-base").model.get_audio_embedding): the pieces that are not GEMM /
 * LayerNorm / GELU.  patchify: log-mel x [batch][frames][mel] -> BatchNorm over mel bins ((x - mean) * scale + bias),
 * 4-tap bicubic stretch of the time axis (tap_idx / tap_w [spec * spec / mel][4], host built), time-chunk fold to a
 * spec x spec image, patch x patch im2col -> bf16 planes [batch * (spec/patch)^2][ldo] (k = kh * patch + kw).
 * window_attn: Swin attention over 8x8 windows of a [batch][H][W] token map, qkv fp32 rows [q | k | v] of 3C in token
 * order; cyclic shift, relative-position bias (table [(2*8-1)^2][heads]) and the shifted-window mask (-100) are applied
 * in-kernel; context written to bf16 planes at the same token rows.  patch_merge: 2x2 neighbourhood gather to 4C-wide
 * rows (quadrant order (0,0),(1,0),(0,1),(1,1)).  mean_rows: per-clip mean over L token rows.  l2_normalize_rows in place
 * (x / max(||x||, eps)).
 * ------------------------------------------------------------------------------------------- */
/* waveform [batch][n] (48 kHz) -> log-mel dB [batch][n/480 + 1][64]: STFT n_fft 1024 / hop 480 / center-reflect with `window`
 * [1024], radix-2 FFT with `twiddle` [512] (cos, -sin) pairs of exp(-2 pi i k / 1024), power, mel filters melw [64][513]
 * (non-zero over bins [mel_lo, mel_hi)), 10 log10(max(x, 1e-10)).  quantize_int16 != 0 applies laion_clap's
 * int16_to_float32(float32_to_int16(x)) round trip (clap_embeddings.py:139) to each sample as it is read. */
int llark_clap_logmel(const float* wav, int batch, int n, int quantize_int16, const float* window, const float* twiddle,
                      const float* melw, const int* mel_lo, const int* mel_hi, float* out, llark_stream_t stream);
int llark_clap_patchify(const float* x, int batch, int frames, int mel, const float* bn_mean, const float* bn_scale,
                        const float* bn_bias, const int* tap_idx, const float* tap_w, int spec, int patch, void* out_hi,
                        void* out_lo, int ldo, llark_stream_t stream);
int llark_clap_window_attn(const float* qkv, int ldq, int batch, int H, int W, int C, int heads, int window, int shift,
                           const float* bias_table, void* out_hi, void* out_lo, void* out_hi_dup, int ldo, llark_stream_t stream);
/* fp32-class linears over fp32 checkpoint weights W = W_hi + W_lo as ONE launch: the producer writes the activation as a
 * K-concatenated [hi | lo | hi] operand (out_hi_dup = second copy of the hi plane, same row stride) and the product runs
 * non-split against [W_hi | W_hi | W_lo].  layernorm_bf16_dup / window_attn (above) / gemm16_act are the producers;
 * gemm16_act is a GEMM whose epilogue is acc + bias -> optional exact (erf) GELU (act = 2) -> 16-bit planes
 * (epilogue LLARK_EPI_SPLIT16 or LLARK_EPI_OUT16), i.e. fc1 + GELU of a transformer MLP without the fp32 round trip. */
int llark_layernorm_bf16_dup(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta, float eps,
                             void* out_hi, void* out_lo, void* out_hi_dup, int ldo, llark_stream_t stream);
int llark_gemm16_act(int variant, int dtype, int split, int epilogue, const void* a_hi, const void* a_lo, int lda, const void* wt,
                     int ldw, const float* bias, int m, int n, int kp, void* out_hi, void* out_lo, void* out_hi_dup, int ldo,
                     int act, llark_stream_t stream);
int llark_clap_patch_merge(const float* x, int ldx, int batch, int H, int W, int C, float* out, int ldo, llark_stream_t stream);
int llark_mean_rows_f32(const float* x, int ldx, int batch, int L, int C, float* out, int ldo, llark_stream_t stream);
int llark_relu_split_bf16(const float* x, int ldx, int rows, int width, void* out_hi, void* out_lo, int ldo, llark_stream_t stream);
int llark_l2_normalize_rows(float* x, int ldx, int rows, int width, float eps, llark_stream_t stream);

/* Decode step, fused: RoPE of the new token (qkv fp32 [batch][3 * nh * 128]) + append to the KV caches + attention over the
 * cache, one launch (m2t/models/llamav2.py:339-365 decode loop; replaces rope_split_heads + attn_decode).  Position = pos,
 * or *pos_dev when pos_dev != NULL.  lo planes: all or none.  alibi_slopes may be NULL. */
int llark_attn_decode_rope_bf16(const float* qkv, int batch, int nh, int hd, int pos, const int* pos_dev, const float* cos_t,
                                const float* sin_t, int max_pos, void* k_cache, void* vt_cache, void* k_cache_lo, void* vt_cache_lo,
                                int smax, void* out, void* out_lo, const float* alibi_slopes, llark_stream_t stream);
/* Decode-step forms with the sequence position in DEVICE memory (*pos_dev = tokens already cached = position of the
 * new token; s = 1): lets ONE captured hipGraph of the whole decode step serve every generated token of
 * m2t/models/llamav2.py:339-365 / m2t/infer.py:137-148. */
int llark_rope_split_heads_dpos(const float* qkv, int batch, int nh, int hd, const int* pos_dev, const float* cos_t,
                                const float* sin_t, void* q, void* k_cache, void* vt_cache, void* q_lo, void* k_cache_lo,
                                void* vt_cache_lo, int smax, llark_stream_t stream);
int llark_attn_decode_bf16_dpos(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                                const void* k_cache_lo, const void* vt_cache_lo, int batch, int nh, int hd,
                                const int* pos_dev, int smax, void* out, void* out_lo, llark_stream_t stream);
/* CrossEntropyLoss on shifted logits (m2t/models/llamav2.py:316-325). logits fp32 [batch*s][ldl]; labels
 * int64 [batch][s]; row_loss scratch float[batch*s]; loss_out float[2] = {mean loss, counted rows}. */
int llark_cross_entropy_shifted(const float* logits, int ldl, int batch, int s, int vocab, const int64_t* labels,
                                int64_t ignore_index, float* row_loss, float* loss_out, llark_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training step of the LLM half: backward / optimizer pieces around the GEMMs (m2t/train.py:53-277 -> HF Trainer:
 * WrappedLlamav2ForCausalLM.forward(labels), loss.backward(), AdamW; scripts/training/train_llark.sh:20-49).
 * ------------------------------------------------------------------------------------------- */
/* dst[b][c][r] = src[b][r][c] (16-bit), dst columns [rows, ld_dst) zero-filled. */
int llark_transpose16(const void* src, int ld_src, int rows, int cols, void* dst, int ld_dst, int batch,
                      long long stride_src, long long stride_dst, llark_stream_t stream);
/* x [batch*s][nh*hd] (16-bit) -> y [batch][nh][s][hd] */
int llark_split_heads16(const void* x, int batch, int s, int nh, int hd, void* y, llark_stream_t stream);
/* P[b][i][j] = softmax_j(scale*scores[b][i][j]), j <= i (bf16, pitch ldp, zero elsewhere). scores fp32 [batch][s][s]. */
int llark_causal_softmax_rows(const float* scores, int batch, int s, float scale, void* p_out, int ldp, llark_stream_t stream);
/* softmax backward: dS = P o (dP - rowsum(P o dP)) * scale  (bf16, pitch ldp) */
int llark_attn_ds(const void* p, const float* dp, int batch, int s, float scale, void* ds_out, int ldp, llark_stream_t stream);
/* RoPE backward + merge heads: dq/dk/dv fp32 [batch][nh][s][hd] -> dqkv bf16 [batch*s][3*nh*hd] */
int llark_rope_merge_bwd(const float* dq, const float* dk, const float* dv, const float* cos_t, const float* sin_t, int batch,
                         int s, int nh, int hd, int pos0, void* dqkv, llark_stream_t stream);
/* LlamaRMSNorm backward: dx (=|+=) ..., dw += ... (fp32 atomics) */
int llark_rmsnorm_bwd(const float* x, const float* w, const float* dy, int rows, int width, float eps, float* dx, int accumulate,
                      float* dw, llark_stream_t stream);
/* ... that also writes the final dx as bf16 [rows][ld16] (ld16 >= width, a multiple of 4): the A operand of the products that follow in the
 * training step, without a llark_split16 pass over the fp32 tensor (same rounding as its hi plane). */
int llark_rmsnorm_bwd_out16(const float* x, const float* w, const float* dy, int rows, int width, float eps, float* dx, int accumulate,
                            float* dw, void* dx16, int ld16, llark_stream_t stream);
/* SwiGLU on the interleaved [32 gate | 32 up] layout: gu fp32 [rows][2*inter] */
int llark_swiglu_fwd(const float* gu, int rows, int inter, void* act, llark_stream_t stream);
int llark_swiglu_bwd(const float* gu, const float* dact, int rows, int inter, void* dgu, llark_stream_t stream);
/* out (one double in device memory) (= | +=) sum_i x[i]^2 over n contiguous fp32 values: the squared global gradient norm behind HF
 * Trainer's clip_grad_norm_ (transformers TrainingArguments.max_grad_norm = 1.0, not overridden by scripts/training/train_llark.sh;
 * reached from m2t/train.py:53-277 -> trainer.train()).  accumulate == 0 zeroes `out` on the stream first; != 0 adds to it, so the
 * norm can be collected slice by slice while the backward is still running. */
int llark_sumsq_f32(const float* x, long long n, double* out, int accumulate, llark_stream_t stream);
/* d(mean shifted CE)/dlogits * loss_scale as bf16 [batch*s][ldd]; row_loss / loss_cnt come from llark_cross_entropy_shifted */
int llark_cross_entropy_bwd(const float* logits, int ldl, int batch, int s, int vocab, const int64_t* labels,
                            const float* row_loss, const float* loss_cnt, float loss_scale, void* dlogits, int ldd,
                            llark_stream_t stream);
int llark_colsum_f32(const float* x, int ld, int rows, int cols, float* out, llark_stream_t stream);
int llark_gather_rows_f32(const float* src, int ld_src, const int64_t* idx, int n, int cols, float* dst, int ld_dst,
                          llark_stream_t stream);
int llark_scatter_add_rows_f32(const float* src, int ld_src, const int64_t* idx, int n, int cols, float* dst, int ld_dst,
                               llark_stream_t stream);
/* torch.optim.AdamW step on bf16 (param_dtype 1) or fp32 (2) parameters with fp32 gradients (scaled by grad_scale) and moments */
int llark_adamw(int param_dtype, void* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int step, float grad_scale, llark_stream_t stream);
/* the same step with HF Trainer's clip_grad_norm_ (TrainingArguments.max_grad_norm) applied on the device: grad_sumsq = the squared norm
 * of the whole unscaled gradient (llark_sumsq_f32, device memory); gradients are multiplied by
 * grad_scale * min(1, max_grad_norm / (sqrt(*grad_sumsq) * grad_scale + 1e-6)) -- no host read between backward and update. */
int llark_adamw_clip(int param_dtype, void* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float grad_scale, const double* grad_sumsq, float max_grad_norm,
                     llark_stream_t stream);
/* Decode-step launches that overlap across their boundaries (round 6; model.generate -> LlamaModel.forward with ONE new token,
 * m2t/infer.py:146, m2t/models/llamav2.py:339-365): a chained consumer is launched on ANOTHER stream while its producer still runs; it fills
 * its LDS ring with weights (64 KiB per workgroup: it fits a CU next to the producer's workgroup), waits until the producer's monotonic
 * arrival counter *wait has reached wait_target before it reads activations / the residual, and adds 1 per workgroup to *signal when its own
 * outputs are written and released.  Counters are caller-owned device words (zeroed once); a producer launch advances its counter by its
 * grid size: llark_gemv16_dma_blocks(epilogue, n) for the Linears, nh * batch for the attention.  The spin is bounded (a lost producer gives
 * wrong data, never a hung GPU).  Results equal the unchained entry points (llark_gemv16_dma, llark_gemv16_dma_rmsnorm,
 * llark_attn_decode_rope_bf16).  llark_gemv16_dma_chain: x != NULL selects the RMSNorm-fused form; m == 1, kp <= 4096. */
int llark_gemv16_dma_blocks(int epilogue, int n);
int llark_gemv16_dma_chain(int split, int epilogue, const void* a_hi, const void* a_lo, int lda, const float* x, int ldx, const float* norm_w,
                           float eps, const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c, int ldc, const float* resid,
                           int ldr, void* out_hi, void* out_lo, int ldo, const unsigned* wait, unsigned wait_target, unsigned* signal,
                           llark_stream_t stream);
int llark_attn_decode_rope_bf16_chain(const float* qkv, int batch, int nh, int hd, int pos, const int* pos_dev, const float* cos_t,
                                      const float* sin_t, int max_pos, void* k_cache, void* vt_cache, void* k_cache_lo, void* vt_cache_lo,
                                      int smax, void* out, void* out_lo, const float* alibi_slopes, unsigned* done, llark_stream_t stream);
/* Layout glue of the Llama training step folded into the kernels around it (round 6; torch autograd of LlamaAttention.forward under
 * WrappedLlamav2ForCausalLM.forward + loss.backward(), m2t/models/llamav2.py:259-337, m2t/train.py:53-277):
 *   llark_gemm16_fragw_rope_qkv_train: llark_gemm16_fragw_rope_qkv in the plain bf16 mode that ALSO writes V row-major,
 *     v_rm [batch][nh][s][128] (the attention backward's operand; no llark_transpose16 of the V^T cache);
 *   llark_attn_backward_bf16_fused: llark_attn_backward_bf16 that reads dO TOKEN-major ([batch*s][ld_do], head h at column 128 h: what the
 *     o_proj dX product leaves; no llark_split_heads16) and writes d(q | k | v) of the fused projection as bf16 dqkv [batch*s][3 nh 128] with
 *     the RoPE backward applied to the q and k parts (bit-equal to llark_rope_merge_bwd over the fp32 outputs of llark_attn_backward_bf16). */
int llark_gemm16_fragw_rope_qkv_train(const void* a, int lda, const void* wfrag, int kp, int batch, int s, int nh, int hd, int pos0,
                                      const float* cos_t, const float* sin_t, int max_pos, void* q, void* k_cache, void* vt_cache, int smax,
                                      void* v_rm, llark_stream_t stream);
int llark_attn_backward_bf16_fused(const void* q, const void* k_cache, const void* v_rm, const void* dO, int ld_do, const void* o,
                                   const float* lse, float* dsum, int batch, int s, int nh, int hd, int smax, const float* cos_t,
                                   const float* sin_t, int pos0, int max_pos, void* dqkv, llark_stream_t stream);
/* dW = dY^T . X without transposed copies of dY (csrc/gemm_bda.hip, round 6; torch autograd of nn.Linear: grad_weight = grad_output^T .
 * input, under WrappedLlamav2ForCausalLM.forward + loss.backward(), m2t/models/llamav2.py:259-337, m2t/train.py:53-277):
 *   llark_pack_frag_t16: src [kp][ld] 16-bit (row = contraction index: the token; column = feature) -> dst = the fragment-major copy of
 *     src^T ([n][kp]; the layout of llark_pack_weight16_frag), kp % 64 == 0, n % 8 == 0; dst holds round_up(n, 32) * kp elements;
 *   llark_gemm16_ta_fragw: c[m][n] (= | +=) sum_k a[k][m] . B(n, k) with a [kp][lda] bf16 contraction-major (dY as it stands) and wfrag =
 *     B fragment-major (llark_pack_frag_t16 of X).  epilogue LLARK_EPI_F32 / LLARK_EPI_RESID (resid may alias c); sumsq (nullable) += sum of
 *     squares of every stored value (as llark_gemm16_t_sumsq).  m % 8 == 0, kp >= 192; LLARK_ERR_UNSUPPORTED otherwise. */
int llark_pack_frag_t16(const void* src, int ld, int kp, int n, void* dst, llark_stream_t stream);
int llark_gemm16_ta_fragw(int epilogue, const void* a, int lda, const void* wfrag, int m, int n, int kp, float* c, int ldc,
                          const float* resid, int ldr, double* sumsq, llark_stream_t stream);
/* The same product on the 16x16x32 MFMA shape (csrc/gemm_bda16.hip, round 6): per joule that instruction does more, and the training
 * GEMMs run against the power limit.  llark_pack_frag_t16x16 = llark_pack_frag_t16 in 16-row chunks (chunk (feature / 16, token / 32), lane
 * l = the 8 tokens 32 q + 8 (l / 16) .. of feature 16 R + l % 16; same extent); llark_gemm16_ta_fragw16 = llark_gemm16_ta_fragw over it.
 * Additionally kp % 128 == 0, kp >= 256, n / ldc / ldr multiples of 4, c / resid 16-byte aligned: LLARK_ERR_UNSUPPORTED otherwise (the
 * caller keeps llark_gemm16_ta_fragw).  32 products per MFMA: agrees with llark_gemm16_ta_fragw to fp32 rounding, not bit for bit. */
int llark_pack_frag_t16x16(const void* src, int ld, int kp, int n, void* dst, llark_stream_t stream);
int llark_gemm16_ta_fragw16(int epilogue, const void* a, int lda, const void* wfrag, int m, int n, int kp, float* c, int ldc,
                            const float* resid, int ldr, double* sumsq, llark_stream_t stream);
/* The two SwiGLU products of the training step with the element-wise pass in their epilogues (csrc/gemm_bda.hip; HF LlamaMLP.forward and
 * its autograd under WrappedLlamav2ForCausalLM.forward + loss.backward(), m2t/models/llamav2.py:224-234,259-337, m2t/train.py:53-277).
 * Plain bf16 operands, fragment-major weights (llark_pack_weight16_frag / llark_adamw_twins).  gu16 [m][ldg >= 2 I] = the gate | up
 * pre-activations as bf16, interleaved like the fused gate/up weight rows ([32 gate | 32 up] per 64 columns).
 *   mode 0: a = x [m][kp], wfrag = gate|up twin (n = 2 I): out = act [m][ldo >= I] = silu(gate) * up; gu16 is WRITTEN.
 *   mode 1: a = d(h) [m][kp], wfrag = down_proj^T twin (n = I): out = d(gate | up) [m][ldo >= 2 I]; gu16 is READ.
 * LLARK_ERR_UNSUPPORTED when the DMA loop does not take the shape (kp < 192, operand beyond 2 GiB). */
int llark_gemm16_fragw_swiglu_train(int mode, const void* a, int lda, const void* wfrag, int m, int n, int kp, void* out, int ldo,
                                    void* gu16, int ldg, llark_stream_t stream);
/* The bf16 step of llark_adamw / llark_adamw_clip (grad_sumsq nullable = no clipping) on a weight MATRIX p [n][k] that also writes the
 * fragment-major operand twins of the UPDATED weight (what the reference reaches through HF Trainer's optimizer.step(), m2t/train.py:255-260,
 * followed by the next micro-batch's nn.Linear forward / backward, m2t/models/llamav2.py:259-337): wfrag (nullable) =
 * llark_pack_weight16_frag(p), the B operand of the forward product -- rows below rope_rows (the q and k parts of a fused q|k|v weight)
 * in llark_gemm16_fragw_rope_qkv's head-permuted order; wtfrag (nullable) = llark_pack_weight16_frag of p^T ([k][n]), the B operand
 * of dX = dY . W.  n % 32 == 0, k % 128 == 0, wtfrag needs n % 64 == 0, rope_rows % 128 == 0, every pointer 16-byte aligned.
 * Parameters and moments are bit-equal to llark_adamw / llark_adamw_clip. */
int llark_adamw_twins(void* p, const float* g, float* m, float* v, int n, int k, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int step, float grad_scale, const double* grad_sumsq, float max_grad_norm, void* wfrag,
                      int rope_rows, void* wtfrag, llark_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LLARK_HIP_H */
