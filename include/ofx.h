/* libofx_hip.so — C ABI of the MI355X-native OutfitX compatibility-scoring forward path.
 *
 * The reference (Krual-T/OutfitX) is 100 % Python and has NO FFI / plugin interface; its boundary
 * for this path is the Python nn.Module API `src.models.OutfitX` (reference src/models/outfit_x.py).
 * This header is the boundary that sits directly UNDER that API: every entry point names the
 * reference symbol whose arithmetic it replaces.  The Python host (outfitx_amd/) binds it with
 * ctypes; see INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only (no torch types); every pointer is a DEVICE pointer
 * unless marked host; every launch goes to the hipStream_t passed as `stream` (void*); no function
 * allocates caller-visible memory; return 0 on success or a negative OFX_E* code with text in
 * ofx_last_error() (thread-local).  gfx950 only.
 *
 * Host synchronisation and state - the complete list:
 *   - no entry point waits for the stream it launches on, with ONE bounded exception: ofx_clip_preprocess /
 *     ofx_vit_b32_fwd_u8 copy a few KB of host-computed resampling plan through a library-owned ring of 4 pinned
 *     slots per device and wait (hipEventSynchronize) only if the copy issued 4 calls earlier on that device has not
 *     left its slot yet;
 *   - ofx_profile_read waits for its events (it is a read-back; profiling is off by default);
 *   - library-owned state: the ofx_handle (packed weight arenas on the handle's device), the per-device pinned ring
 *     above, a host-side cache of resampling coefficient tables, the ofx_tune knobs and the profiling records.
 *     Nothing else is process-global; handles of different devices are independent.
 */
#ifndef OFX_H
#define OFX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFX_ABI_VERSION 5

enum { OFX_OK = 0, OFX_EINVAL = -1, OFX_ESHAPE = -2, OFX_EHIP = -3, OFX_EWORKSPACE = -4, OFX_ESTATE = -5 };
enum ofx_dtype { OFX_F32 = 0, OFX_BF16 = 1, OFX_F16 = 2 };
/* OFX_ACT_MISH_GRAD (backward): out = acc * mish'(resid) - `resid` then carries the saved pre-activation, not an addend */
enum ofx_act { OFX_ACT_NONE = 0, OFX_ACT_QUICK_GELU = 1, OFX_ACT_GELU = 2, OFX_ACT_MISH = 3, OFX_ACT_MISH_GRAD = 4 };
/* MFMA operand precision of a sub-model.  BF16X3 = three bf16 products per term
 * (hi*hi + lo*hi + hi*lo, realised as one K-concatenated GEMM) ~ fp32-grade results. */
/* OFX_PREC_F16W2 (outfit transformer only, round 4): f16 activations against split (hi, lo) f16 weights - two MFMA products per weight on ONE copy of
 * the activations (the towers' scheme): two thirds of bf16x3's weight bytes and matrix work, ~1e-4 instead of ~1e-5 from the fp32 reference; scoring only. */
enum ofx_precision { OFX_PREC_BF16 = 0, OFX_PREC_F16 = 1, OFX_PREC_BF16X3 = 2, OFX_PREC_F16W2 = 3 };
enum ofx_out_kind { OFX_OUT_F32 = 0, OFX_OUT_OP = 1, OFX_OUT_SPLIT3 = 2 };

typedef struct ofx_handle ofx_handle;
typedef void* ofx_stream; /* hipStream_t */

const char* ofx_last_error(void);
int ofx_abi_version(void);

/* ------------------------------------------------------------------ model description ------ */
typedef struct ofx_model_desc {
    /* global outfit Transformer — reference src/models/configs/transformer_config.py:8-24 */
    int d_model, n_head, d_ffn, n_layers, max_items; /* 1024, 16, 2024, 6, 16 */
    int outfit_act;                                   /* OFX_ACT_MISH */
    int outfit_precision;                             /* ofx_precision */
    /* CLIP ViT-B/32 vision tower — HF CLIPVisionConfig as loaded by clip_image_encoder.py:20-22 */
    int vit_width, vit_layers, vit_heads, vit_mlp, vit_patch, vit_image, vit_act; /* 768,12,12,3072,32,224 */
    /* CLIP text tower — HF CLIPTextConfig as loaded by clip_text_encoder.py:19-21 */
    int txt_width, txt_layers, txt_heads, txt_mlp, txt_vocab, txt_max_pos, txt_act, txt_eos_id; /* 512,12,8,2048,49408,77 */
    int proj_dim;                                     /* 512 */
    int tower_precision;                              /* OFX_PREC_BF16 | OFX_PREC_F16: operand type of both towers */
    float ln_eps;                                     /* 1e-5 */
    /* Operand scheme of the towers beyond one product per term (all zero = single product everywhere):
     * vit_w2_mask: OFX_W2_* bits - these ViT GEMMs multiply against split (hi, lo) weights, two MFMA products per weight;
     * txt_x3:      the text tower runs three products per term (hi*hi + lo*hi + hi*lo, K-concatenated);
     * proj_x3:     the ViT's post-LayerNorm + visual_projection tail runs three products per term;
     * vit_x3:      the whole ViT runs three products per term (fp32 residual stream, materialised LayerNorms; the MFMA attention core
     *              stays on once-rounded q, k, v, P): ~1.5e-4 end to end on any weight draw at 1.9x the time of the default scheme;
     *              vit_w2_mask then only matters for the patch embedding. */
    int vit_w2_mask, txt_x3, proj_x3, vit_x3;
    /* Finer rungs of the same ladder (round 4; all zero = the behaviour above):
     * txt_w2_mask: OFX_W2_QKV / OUT / FC1 / FC2 bits - with txt_x3 = 0 the TEXT tower runs the ViT's scheme: f16 activations against split
     *              (hi, lo) weights on those GEMMs (1.5-2 product equivalents instead of 3), MFMA attention on once-rounded q, k, v;
     *              its final LayerNorm + text_projection stay three-product when proj_x3 is set;
     * vit_w2_qkv_layers / vit_w2_fc1_layers / vit_w2_out_layers / vit_w2_fc2_layers: bit l set = ViT layer l's qkv / fc1 / out-proj / fc2 GEMM
     *              takes the split weights that the OFX_W2_* bit asks for; 0 = every layer.  (The other layers run the single-product copy:
     *              qkv through the fused QKV + attention kernel.) */
    int txt_w2_mask, vit_w2_qkv_layers, vit_w2_fc1_layers, vit_w2_out_layers, vit_w2_fc2_layers;
} ofx_model_desc;
enum { OFX_W2_PATCH = 1, OFX_W2_QKV = 2, OFX_W2_OUT = 4, OFX_W2_FC1 = 8, OFX_W2_FC2 = 16 };

/* Fills *d with the configuration of the reference's type='clip' model. */
void ofx_default_desc(ofx_model_desc* d);

ofx_handle* ofx_create(int device, const ofx_model_desc* desc);
void ofx_destroy(ofx_handle* h);

/* ------------------------------------------------------------------ weight packing --------- *
 * fp32 torch parameters (device pointers, contiguous) -> MFMA operand layout in the handle's own
 * HBM arena.  Re-run after the parameters change (optimizer.step / load_state_dict).
 * Pointer order = state_dict order of the reference module:
 *   outfit: [outfit_token, target_item_image_emb, cp_ffn.1.weight, cp_ffn.1.bias, cir_ffn.0.weight]
 *           then per layer: in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias,
 *           linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.w, norm1.b, norm2.w, norm2.b
 *   vision: [class_embedding, patch_embedding.weight, position_embedding.weight, pre_layrnorm.w, .b]
 *           then per layer: k_proj.w,.b, v_proj.w,.b, q_proj.w,.b, out_proj.w,.b, layer_norm1.w,.b,
 *           mlp.fc1.w,.b, mlp.fc2.w,.b, layer_norm2.w,.b ; then post_layernorm.w,.b, visual_projection.weight
 *   text:   [token_embedding.weight, position_embedding.weight] then per layer (as vision) ;
 *           then final_layer_norm.w,.b, text_projection.weight                                  */
int ofx_pack_outfit_weights(ofx_handle* h, const void* const* fp32_params, int n_params, ofx_stream stream);
int ofx_pack_vision_weights(ofx_handle* h, const void* const* fp32_params, int n_params, ofx_stream stream);
int ofx_pack_text_weights(ofx_handle* h, const void* const* fp32_params, int n_params, ofx_stream stream);

/* ------------------------------------------------------------------ workspace -------------- */
enum ofx_op { OFX_OP_SET_ENCODER = 0, OFX_OP_VIT = 1, OFX_OP_TEXT = 2, OFX_OP_TOPK = 3 };
/* bytes of scratch the op needs for `n` units (outfits / images / texts / queries) of `len`
 * (max items incl. none / unused / tokens per text / pool rows). */
size_t ofx_workspace_bytes(ofx_handle* h, int op, int n, int len);

/* ------------------------------------------------------------------ hot path --------------- */
/* Rows B/C/D of SURVEY §8a: prefix-token concat + key-padding mask + the 6-layer pre-norm
 * Transformer encoder + "take row 0" — replaces src/models/outfit_x.py:129-142 (CP) and
 * :154-168 (CIR) incl. the nn.TransformerEncoder call.  Padded items are skipped (pad-free).
 *   x         [B, L, d_model] fp32 item embeddings (row-major)
 *   pad_mask  [B, L] uint8/bool, non-zero = padded item (reference: True = pad)
 *   prefix    [d_model] (prefix_stride 0: CP outfit_token) or [B, d_model] (stride d_model: CIR)
 *             fp32; NULL = use the packed outfit_token
 *   out_row0  [B, d_model] fp32: encoder output at the prefix position                         */
int ofx_set_encoder_fwd(ofx_handle* h, const float* x, const uint8_t* pad_mask, const float* prefix,
                        int prefix_stride, int B, int L, float* out_row0, void* ws, size_t ws_bytes,
                        ofx_stream stream);
/* cp_ffn = Dropout -> Linear(d_model,1): raw logit per outfit (outfit_x.py:57-61,143). */
int ofx_cp_head(ofx_handle* h, const float* row0, int B, float* logits, ofx_stream stream);
/* cir_ffn = Linear(d_model,d_model,bias=False) (outfit_x.py:65-67,171). */
int ofx_cir_head(ofx_handle* h, const float* row0, int B, float* emb, void* ws, size_t ws_bytes, ofx_stream stream);
/* CIR prefix token [target_item_image_emb ‖ target text emb] (outfit_x.py:154-158): out [B,d_model]. */
int ofx_cir_prefix(ofx_handle* h, const float* target_text_emb, int B, float* prefix_out, ofx_stream stream);

/* Row F: HF CLIPVisionModelWithProjection(pixel_values).image_embeds (clip_image_encoder.py:74-76)
 * + optional F.normalize (base_image_encoder.py:46-47).  pixels [N,3,224,224] fp32 (already
 * resized/normalised by the host processor); emb [N, emb_ld] fp32, written at column emb_col.   */
int ofx_vit_b32_fwd(ofx_handle* h, const float* pixels, int N, float* emb, int emb_ld, int emb_col,
                    int normalize, void* ws, size_t ws_bytes, ofx_stream stream);
/* Row G: HF CLIPTextModelWithProjection(input_ids, attention_mask).text_embeds
 * (clip_text_encoder.py:56-58) + optional F.normalize.  ids/mask [N,T] int64.  `lengths` is a HOST
 * array [N] of tokens to compute per text (= EOS position + 1; causal attention makes later tokens
 * irrelevant), or NULL to compute all T.                                                        */
int ofx_clip_text_fwd(ofx_handle* h, const int64_t* ids, const int64_t* attn_mask, const int* lengths_host,
                      int N, int T, float* emb, int emb_ld, int emb_col, int normalize, void* ws,
                      size_t ws_bytes, ofx_stream stream);

/* Row H: torch.cdist(y[B,1,D], cand[B,C,D]).squeeze(1).argmin(-1) (fill_in_the_blank_trainer.py:50-56).
 * idx int64 [B]; dist fp32 [B,C] optional.                                                      */
int ofx_fitb_argmin(const float* y_hat, const float* cand, int B, int C, int D, int64_t* idx, float* dist,
                    ofx_stream stream);
/* Row I: torch.cdist(Q,P) -> topk(k, largest=False) (complementary_item_retrieval_trainer.py:240-249).
 * fp32-exact distances; ascending; ties -> smaller pool index.  idx int64 [nq,k] = row + index_base.  */
int ofx_l2_topk(ofx_handle* h, const float* Q, const float* P, int nq, int np, int D, int k, int64_t index_base,
                int64_t* idx, float* dist, void* ws, size_t ws_bytes, ofx_stream stream);
/* Merge `parts` candidate lists (after the RCCL all-gather of per-shard top-k): in [parts,nq,k]. */
int ofx_topk_merge(const int64_t* idx_in, const float* dist_in, int parts, int nq, int k, int64_t* idx, float* dist,
                   ofx_stream stream);

/* ------------------------------------------------------------------ image preprocessing ("next" row N2) --- */
/* CLIPImageProcessor(do_convert_rgb=False) as the reference runs it on the host (clip_image_encoder.py:29-31,69-71): resize the
 * shortest edge to `size` with PIL's antialiased BICUBIC (ImagingResample: 22-bit fixed-point coefficients, horizontal then
 * vertical, 8-bit intermediate — bit-exact), centre crop size x size, x 1/255, (x - mean) / std -> out [N, 3, size, size] fp32
 * (device), equal to the host pipeline bit for bit.  src: device buffer holding the N uint8 images row-major with `channels`
 * (3 = RGB interleaved, 1 = grey, replicated) at byte offsets[i]; offsets / heights / widths are HOST arrays.  RGB pixels are
 * fetched as unaligned dwords: src must stay readable 1 byte past the end of every image (give the buffer >= 4 bytes of slack). */
size_t ofx_clip_preprocess_ws(const int* heights, const int* widths, int N, int channels, int size);
int ofx_clip_preprocess(const uint8_t* src, const long long* offsets, const int* heights, const int* widths, int N, int channels, int size,
                        const float* mean, const float* stdv, float* out, void* ws, size_t ws_bytes, ofx_stream stream);
/* The two fused: decoded uint8 images -> image embeddings (clip_image_encoder.py:66-76 in one call).  The preprocessor's
 * vertical pass writes the operand-type im2col rows of the patch-embedding GEMM directly, so neither the fp32 pixel_values
 * tensor (0.6 MB / image) nor the patchify pass exists; results equal ofx_clip_preprocess + ofx_vit_b32_fwd bit for bit.
 * mean / stdv: the processor's image_mean / image_std (host, 3 floats); size and patch come from the model descriptor. */
size_t ofx_vit_b32_u8_ws_bytes(ofx_handle* h, const int* heights, const int* widths, int N, int channels);
int ofx_vit_b32_fwd_u8(ofx_handle* h, const uint8_t* src, const long long* offsets, const int* heights, const int* widths, int N, int channels,
                       const float* mean, const float* stdv, float* emb, int emb_ld, int emb_col, int normalize, void* ws, size_t ws_bytes,
                       ofx_stream stream);

/* ------------------------------------------------------------------ indexed (varlen) set input ("next" row N3) --- */
/* The same encoder with the outfits given as ROW INDICES into a device-resident embedding table [n_table, ld] fp32
 * (the precomputed-embedding store kept in HBM) instead of a padded [B, L, D] tensor + mask: outfit b holds items
 * item_index[cu_items[b] .. cu_items[b+1]); cu_items[0] = 0; at most max_len (<= 63; <= 31 for the training entry points) items per outfit.  Replaces the
 * reference's collate (outfit_x_base_processor.py:20-81: per-item torch.tensor + cat + stack, then a 12.6 MB/256-outfit
 * H2D copy) by a few KB of indices.  Bit-identical to the dense call on the same items.  Workspace: ofx_workspace_bytes
 * (OFX_OP_SET_ENCODER, B, max_len). */
int ofx_set_encoder_fwd_indexed(ofx_handle* h, const float* table, int ld, long long n_table, const int* item_index, const int* cu_items,
                                const float* prefix, int prefix_stride, int B, int max_len, float* out_row0, void* ws, size_t ws_bytes,
                                ofx_stream stream);

/* ------------------------------------------------------------------ training step ("next" row N1) --- */
/* CP path on precomputed embeddings with a tape, and its backward: what torch autograd computes for the reference's
 * CP trainer step (compatibility_prediction_trainer.py:57-81) with dropout = 0.  Needs outfit_precision BF16 or F16.
 * grads: one fp32 buffer, layout from ofx_cp_train_grad_floats (offsets per packed tensor, padded shapes:
 * linear1 [ffn_pad, D], linear2 [D, ffn_pad], ffn_pad = d_ffn rounded up to 128). */
size_t ofx_cp_train_tape_bytes(ofx_handle* h, int B, int L);
size_t ofx_cp_train_ws_bytes(ofx_handle* h, int B, int L);
size_t ofx_cp_train_grad_floats(ofx_handle* h, size_t* offsets, int n_offsets);
/* dropout_p / seed: torch's train-mode dropout (attention probabilities, dropout1, FFN dropout, dropout2 of every layer and
 * the head's nn.Dropout) with stateless masks = hash(seed, site, row, col); the backward call must repeat both values.
 * Same distribution as torch's, not the same random stream. */
int ofx_cp_train_fwd(ofx_handle* h, const float* x, const uint8_t* pad_mask, int B, int L, float* logits, void* tape, size_t tape_bytes,
                     void* ws, size_t ws_bytes, float dropout_p, unsigned seed, ofx_stream stream);
int ofx_cp_train_bwd(ofx_handle* h, void* tape, size_t tape_bytes, const float* dlogits, int B, int L, float* grads, size_t grad_floats,
                     void* ws, size_t ws_bytes, float dropout_p, unsigned seed, ofx_stream stream);
int ofx_cp_train_fwd_indexed(ofx_handle* h, const float* table, int ld, long long n_table, const int* item_index, const int* cu_items,
                             int B, int max_len, float* logits, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                             float dropout_p, unsigned seed, ofx_stream stream);
/* CIR / FITB path (outfit_x.py:147-172) with a tape and its backward: prefix = [target_item_image_emb | target_text[b]],
 * y [B, d_model] = row0 Wc^T (cir_ffn, no bias, no dropout).  The set is given either padded (x, pad_mask) or indexed
 * (table, ld, n_table, item_index, cu_items) - pass NULL for the form not used.  Gradients land in the same flat buffer
 * (slots: target_item_image_emb, cir_ffn weight, all layer tensors; the CP head's and outfit_token's slots are not written). */
int ofx_cir_train_fwd(ofx_handle* h, const float* x, const uint8_t* pad_mask, const float* table, int ld, long long n_table,
                      const int* item_index, const int* cu_items, const float* target_text, int B, int L, float* y, void* tape,
                      size_t tape_bytes, void* ws, size_t ws_bytes, float dropout_p, unsigned seed, ofx_stream stream);
int ofx_cir_train_bwd(ofx_handle* h, void* tape, size_t tape_bytes, const float* dy, int B, int L, float* grads, size_t grad_floats,
                      void* ws, size_t ws_bytes, float dropout_p, unsigned seed, ofx_stream stream);
/* The same backward passes writing every gradient straight into a caller tensor of the PARAMETER's own shape (linear1 [d_ffn, D],
 * linear2 [D, d_ffn], ...): grad_ptrs = host array of 5 + 12 * n_layers device pointers in pack order (entries of tensors that are
 * not on the path are ignored); accumulate != 0 adds to the existing contents - torch's p.grad += g without the extra kernels. */
int ofx_cp_train_bwd_into(ofx_handle* h, void* tape, size_t tape_bytes, const float* dlogits, int B, int L, float* const* grad_ptrs,
                          int n_ptrs, int accumulate, void* ws, size_t ws_bytes, float dropout_p, unsigned seed, ofx_stream stream);
int ofx_cir_train_bwd_into(ofx_handle* h, void* tape, size_t tape_bytes, const float* dy, int B, int L, float* const* grad_ptrs,
                           int n_ptrs, int accumulate, void* ws, size_t ws_bytes, float dropout_p, unsigned seed, ofx_stream stream);
/* Data-parallel overlap: arm one hipEvent_t per outfit-transformer layer (events[l], n = n_layers; n = 0 disarms) for the NEXT
 * ofx_*_train_bwd* call on this handle only.  That call records events[l] on its stream as soon as the 12 gradient tensors of layer l
 * (Win, bin, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2) are final - layers finish last-to-first - so the host can start that
 * layer's gradient all-reduce on another stream while the backward of the layers below still runs (outfitx_amd/trainer.py). */
int ofx_train_arm_layer_events(ofx_handle* h, void* const* events, int n);
/* out[rows, cols] fp32 = keep-mask / (1 - p) of dropout site `site` (layer l: 4l + {0 attention [B*heads, 32*i + j], 1 dropout1,
 * 2 FFN, 3 dropout2}; 4 * n_layers = head), exactly as the kernels compute it.  Test / debugging aid. */
int ofx_dropout_mask(float dropout_p, unsigned seed, int site, int rows, int cols, float* out, ofx_stream stream);
/* FocalLoss(alpha, gamma, mean) forward and d loss / d logits * upstream (src/losses/focal_loss.py:23-41). loss / dlogits may be NULL. */
int ofx_focal_loss(const float* logits, const float* labels, int B, float alpha, float gamma, float upstream, float* loss, float* dlogits,
                   ofx_stream stream);
/* The same with the reference's `reduction` argument (focal_loss.py:36-41): 1 mean, 2 sum, 0 none.  `per_elem` [B] (required for
 * 'none', optional otherwise) receives the unreduced losses; for 'none' dlogits is the per-element derivative (times upstream) and
 * *loss, if given, their sum. */
int ofx_focal_loss_ex(const float* logits, const float* labels, int B, float alpha, float gamma, float upstream, int reduction, float* loss,
                      float* per_elem, float* dlogits, ofx_stream stream);

/* ------------------------------------------------------------------ profiling --------------- */
/* HIP-event timing of every launch, by category {0 GEMM, 1 norm/embed, 2 attention, 3 other}.
 * enable(mask) clears and starts recording the categories in the bit mask (1 GEMM | 2 norm | 4 attention | 8 other; 0 = off); read() waits for the events (host sync) and returns the
 * summed milliseconds, executed FLOPs (GEMM only) and launch counts; arrays of 4.  Process-global, not thread-safe: benchmarks and tests only. */
void ofx_profile_enable(int on);
int ofx_profile_read(double* ms, double* flops, long long* launches);
/* Per-launch records of the last recording, in launch order (call BEFORE ofx_profile_read, which clears them).  GEMM records carry
 * their shape and kernel: M, N, K (logical depth), kmul (executed K = kmul x K: 2 split weights [hi | lo], 3 three-product
 * K-concatenation) and kind (1 128x128 tile kernel incl. its split-K / 64-row variants, 2 256x256, 3 256x128, 4 256x256 ping-pong,
 * 6 dual-weight 256x256, 7 fused QKV projection + attention); flops = executed FLOPs; bytes = ALGORITHMIC HBM bytes of the launch (every operand
 * read once - A [M, K], the weight rows as stored - and every output / in-place stream element read and written once, as the
 * launch's epilogue is configured).  Returns the count. */
typedef struct ofx_prof_record { int cat, M, N, K, kind, kmul; float ms; double flops; double bytes; } ofx_prof_record;
int ofx_profile_records(ofx_prof_record* out, int cap);

/* Process-wide tuning knobs (benchmarks / tests only).  knob 0: GEMM rasterisation group (row panels per L2 group, default 8);
 * knob 6: 2 (default) folds the CLIP towers' LayerNorms into the neighbouring GEMM epilogues AND keeps their residual stream as an
 * operand-type (hi, lo) pair updated in place (no fp32 stream between the layers), 1 folds with an fp32 stream, 0 materialises them;
 * knob 8: 1 (default) the ViT's last layer computes queries for the CLS rows only, 0 runs the full QKV GEMM;
 * knob 9: bit 0 (default on) / bit 1 (default off): ViT layers with single-product / split (hi, lo) q | k | v weights run the fused
 *         QKV-projection + attention kernel; cleared: the GEMM -> HBM -> attention-kernel pair;
 * knob 10: 1 (default) small-batch outfit-transformer GEMMs (split-K plans) leave their second pass to the consumer kernel (set
 *          attention sums the q | k | v slabs; reduce + LayerNorm in one launch), 0 the separate reduce and LayerNorm launches;
 * knob 12: 1 (default) split-weight GEMMs with an fp8 copy of their lo halves run the fp8 correction product, 0 = the f16 one;
 * knob 13: log2 of the activation scale of that product (default 0 with the e5m2 activation image of round 4; 2 in an e4m3 build, -DOFX_F8_ABF8=0).
 * knob 15: three-product GEMMs: 1 (default) the operand-tiles-loaded-once kernel from 192 tiles on, 2 always, 0 never.
 * knob 14: 1 = the persistent split-weight GEMMs launch the smallest grid that finishes in the same number of rounds (the CUs left alone
 * serve the side stream's kernels); default 0 = one block per CU (the trimmed grid measured 0.4 ms per step slower).
 * knob 11: grid size of the persistent dual-weight GEMM (default -1 = one block per CU of the device, each walking its tiles and
 *          fetching the next tile's first k-steps under the current epilogue), 0 = one block per tile.
 * knob 16: 1 (default) gemm_x3_kernel launches one block per CU walking its tiles, 0 = one block per tile (short-lived blocks).
 * knob 17: 1 (default) ofx_l2_topk on pools of >= 32,768 rows runs sample + filter (the distance matrix is never written), 0 = always
 *          distance matrix + radix select.  Same results either way.
 * knob 18: 1 (default) the split-weight GEMM's f16 outputs (qkv, fc1) are stored straight from the accumulator layout, 0 = through the LDS
 *          transpose of rounds 1-3, 2 = as 1 with whole-line stores (8 rows x 128 B per instruction after a DPP row exchange; measured level
 *          with 1: profiles/r04_epilogue_wide_skew.txt).  Bit-identical results.
 * knob 19: experiment, default 0: start skew of gemm_w2f8_kernel's blocks by XCD (v > 0: odd XCDs start v x ~2,000 cycles late, v < 0:
 *          XCD x starts x |v| x ~2,000 cycles late).  Results unchanged; no setting was faster (same file).
 * knob 20: validation only, default 0: 1 = a three-product ViT (vit_x3) keeps q | k | v in fp32 and runs the fp32 attention kernel instead of the MFMA attention
 *          on operand-rounded q | k | v (several times slower; tools/parity_selfcheck.py uses it for its reference run). */
int ofx_tune(int knob, int value);
/* A counter that every ofx_tune call bumps, and whether per-launch profiling events are being recorded: the host mirror replays a
 * stream-captured forward (outfitx_amd/graphs.py) only while the counter still has the value it had at capture time and no recording
 * is on (events ride on launches; a replayed graph issues none). */
unsigned ofx_config_generation(void);
int ofx_profile_enabled(void);
/* A HIP stream of the LOWEST dispatch priority on `device` (hipStreamCreateWithPriority, non-blocking): the host mirror runs the text tower on it
 * beside the ViT on the caller's stream, so that its workgroups take CUs only when the caller's stream has none ready.  Caller destroys it. */
int ofx_stream_create_low_priority(int device, ofx_stream* out);
int ofx_stream_destroy(ofx_stream stream);
/* Diagnostics: when buf != NULL the big-tile GEMM writes {shader cycles, 100 MHz ticks} of its main loop per block (16 B each). */
void ofx_debug_gemm_clock(void* buf);

/* ------------------------------------------------------------------ op level (tests) ------- */
int ofx_gemm(const void* A, const void* W, void* C, const float* bias, const float* resid, int M, int N, int K,
             int lda, int ldc, int ldr, int act, int out_kind, int op_dtype, ofx_stream stream);
/* The same against a SPLIT weight matrix W2 [N, 2K], row n = [hi(K) | lo(K)] (ofx_convert mode 3): C = A (hi + lo)^T + ... with one
 * copy of A - two MFMA products per weight, ~22 significant weight bits.  K multiple of 64. */
int ofx_gemm_w2(const void* A, const void* W2, void* C, const float* bias, const float* resid, int M, int N, int K,
                int lda, int ldc, int ldr, int act, int out_kind, int op_dtype, ofx_stream stream);
/* Three-product GEMM: A3 [M, lda >= 3K] rows [hi | lo | hi], W3 [N, 3K] rows [hi | hi | lo] (ofx_convert mode 2), C = hi.hi + lo.hi + hi.lo.
 * From 192 tiles of 256 x 128 on (ofx_tune(15, 2): always; 0: never) the kernel that stages each operand tile once (gemm_x3.hip), else the
 * K-concatenated single-product kernels on K' = 3K.  K multiple of 32, N of 128. */
int ofx_gemm_x3(const void* A3, const void* W3, void* C, const float* bias, const float* resid, int M, int N, int K,
                int lda, int ldc, int ldr, int act, int out_kind, int op_dtype, ofx_stream stream);
/* The same with the correction product A . lo^T on the block-scaled fp8 matrix instruction (2x the f16 rate; f16 operands only):
 * ofx_pack_lo8 turns the lo halves of W2 [N, 2K] into W8 [N, K] e4m3 bytes (per row scaled to max |lo| 2^sw in [128, 256), the
 * 128-blocks k-permuted as the kernel's in-register fp8 activation image is) + scale8 [N] E8M0 bytes; N and K multiples of 128.
 * ofx_gemm_w2f8 = A hi^T + bf8(A 2^shift) fp8(lo 2^sw)^T 2^-(shift + sw): the weight error drops from 2^-12 to ~2^-15 relative, the
 * activation operand stays the one f16 copy.  Since round 4 the in-register activation image is E5M2 (f16's exponent range, 3 significant bits,
 * shift 0): it follows any finite f16 operand - no magnitude precondition on A; values above 57,344 are clamped there (rounds 3's e4m3 image
 * saturated at |a| > 112 and then corrected such a column only in part).  Small problems (fewer than 256 tiles of 256 x 256) run ofx_gemm_w2's path on W2. */
int ofx_pack_lo8(const void* W2, void* W8, void* scale8, int N, int K, ofx_stream stream);
int ofx_gemm_w2f8(const void* A, const void* W2, const void* W8, const void* scale8, void* C, const float* bias, const float* resid, int M, int N, int K,
                  int lda, int ldc, int ldr, int act, int out_kind, ofx_stream stream);
/* Weight-gradient GEMM of the training step: C[M,N] fp32 = sum_k A[k, m] * B[k, n]; A [K, lda] and B [K, ldb] row-major
 * operand-type matrices whose ROW index is contracted (dW = dY^T X without transposed copies).  M, N multiples of 256.
 * k_dev: optional device-side live row count (<= K).  Both operands must be readable up to round_up(K, 64) rows.
 * slab: ofx_gemm_tn_ws(M,N,K) bytes of scratch for the deterministic split-K (NULL = no split). */
size_t ofx_gemm_tn_ws(int M, int N, int K);
int ofx_gemm_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K, const int* k_dev,
                void* slab, size_t slab_bytes, int op_dtype, ofx_stream stream);
/* ofx_gemm with split-K scratch: slab of ofx_gemm_splitk_ws(M,N,K) bytes (0 = the shape is not split) */
size_t ofx_gemm_splitk_ws(int M, int N, int K);
int ofx_gemm_splitk(const void* A, const void* W, void* C, const float* bias, const float* resid, int M, int N, int K,
                    int lda, int ldc, int ldr, int act, int out_kind, int op_dtype, void* slab, size_t slab_bytes, ofx_stream stream);
int ofx_layernorm(const float* x, const int* row_idx, const float* gamma, const float* beta, void* y, int rows,
                  int D, int ldy, int out_kind, int op_dtype, float eps, ofx_stream stream);
int ofx_attention(const void* qkv, void* out, const int64_t* key_mask, int nseq, int seq_len, int n_head, int ld,
                  int ldo, int k_off, int v_off, int mask_ld, int causal, float scale, int op_dtype, ofx_stream stream);
int ofx_set_attention(const float* qkv, void* out, const int* cu_seqlens, int nseq, int n_head, int D, int ldo,
                      int out_kind, int max_len, int only_row0, float scale, int op_dtype, ofx_stream stream);
/* fp32 [rows, cols] -> operand type; mode 0 plain, 1 [hi|lo|hi] (activation split), 2 [hi|hi|lo] (weight split) */
/* Fused QKV projection + scaled-dot-product attention of a CLIP ViT layer (HF CLIPAttention's q/k/v_proj + softmax(q k^T * scale) v,
 * reached from clip_image_encoder.py:74-76), q | k | v staged in LDS only: X [nseq * seq_len, ldx] operand type, Wqkv [3 width, width]
 * (q | k | v rows), bias [3 width]; optional LayerNorm-fold consumer inputs row_stat [rows, 2] (mean, rstd) + col_sum [3 width];
 * out [nseq * seq_len, ldo] operand type (heads concatenated).  seq_len in [33, 64], width = n_head * 64, no mask.  X and out must not alias. */
int ofx_fused_qkv_attention(const void* X, const void* Wqkv, const float* bias, const float* row_stat, const float* col_sum, void* out,
                            int nseq, int seq_len, int width, int n_head, int ldx, int ldo, float scale, int op_dtype, ofx_stream stream);
/* The same against split weights: Wqkv2 [3 width, 2 width], row n = [hi(width) | lo(width)] (ofx_convert mode 3): q | k | v = X . (hi + lo)^T,
 * two MFMA products per weight on one copy of the activations (the default tower scheme 'f16w2x'). */
int ofx_fused_qkv_attention_w2(const void* X, const void* Wqkv2, const float* bias, const float* row_stat, const float* col_sum, void* out,
                               int nseq, int seq_len, int width, int n_head, int ldx, int ldo, float scale, int op_dtype, ofx_stream stream);
/* fp32-arithmetic attention over fixed-length sequences of <= 64 rows (the three-product CLIP text tower): qkv fp32 [nseq * seq_len, 3 D]
 * (q | k | v), HF's causal AND key-padding mask (key_mask [nseq, mask_ld] int64, 0 = ignored, may be NULL); out as ofx_set_attention. */
int ofx_attention_f32(const float* qkv, void* out, const int64_t* key_mask, int nseq, int seq_len, int n_head, int D, int ldo, int out_kind,
                      int mask_ld, int causal, float scale, int op_dtype, ofx_stream stream);
int ofx_convert(const float* src, void* dst, int rows, int cols, int mode, int op_dtype, ofx_stream stream);

#ifdef __cplusplus
}
#endif
#endif
