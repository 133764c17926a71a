/*
 * ptts.h — C ABI of libptts_hip.so: the MI355X-native Parler-TTS generation hot path.
 *
 * The reference (huggingface/parler-tts) has NO native FFI: its boundary is Python-level
 * (SURVEY.md §8(b)). Each entry point below therefore cites the reference *Python* interface it
 * replaces (file:line under /root/reference/parler_tts/); INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add. Plain C: pointers + sizes only, no torch / C++ types.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = error class (PTTS_E_*); ptts_last_error() returns the
 *     message of the last failing call on the calling thread. No C++ exception crosses this boundary.
 *   - all `*_dev` pointers are DEVICE pointers (HIP). Borrowed for the duration of the call only, unless
 *     stated otherwise; the engine copies / re-packs what it keeps.
 *   - `stream` is a hipStream_t passed as void* (NULL = the engine's own stream). Work is enqueued
 *     asynchronously unless the function is documented as synchronising.
 *   - one engine per stream; an engine is not re-entrant (same as the reference model object,
 *     modeling_parler_tts.py:3297 per-call `_cache`), but may be driven from any host thread.
 */
#ifndef PTTS_H_
#define PTTS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTTS_ABI_VERSION 8

enum { PTTS_F32 = 0, PTTS_BF16 = 1 };

enum {
  PTTS_OK = 0,
  PTTS_E_INVALID = -1,   /* bad argument / shape / state (Python wrapper raises ValueError) */
  PTTS_E_HIP = -2,       /* HIP runtime error */
  PTTS_E_MISSING = -3,   /* weights not fully loaded */
  PTTS_E_CAPACITY = -4,  /* batch / context / encoder length exceeds what the engine was created for */
  PTTS_E_UNSUPPORTED = -5
};

typedef struct ptts_engine ptts_engine; /* decoder LM engine: packed weights, KV arena, sampler state, hipGraph */
typedef struct ptts_dac ptts_dac;       /* DAC codes -> waveform engine */
typedef struct ptts_t5 ptts_t5;         /* T5 description encoder (ABI v7) */

/* ParlerTTSDecoderConfig integers (configuration_parler_tts.py:111-172) + engine capacities. */
typedef struct {
  int32_t hidden_size;
  int32_t num_layers;
  int32_t num_heads;     /* query heads; head_dim = hidden_size / num_heads must be 64 */
  int32_t ffn_dim;
  int32_t num_codebooks;
  int32_t vocab_size;    /* LM-head rows per codebook; embedding tables have vocab_size+1 rows (:1353) */
  int32_t max_positions; /* max_position_embeddings */
  int32_t rope;          /* rope_embeddings */
  float rope_theta;
  int32_t pad_token_id, eos_token_id, bos_token_id;
  int32_t dtype;         /* PTTS_F32 (parity mode, BASELINE configs[0] numerics) | PTTS_BF16 (bf16 weights + KV, fp32 accumulate) */
  int32_t max_batch;     /* utterances per call. It also selects the decode-step kernels at create time (query: ptts_state / DESIGN.md §4):
                            <= 4: row-per-wave GEMV step, 4 KV splits; 5..8: GEMV step with 8-utterance register groups, 2 splits; > 8: MFMA
                            strip step (no row-major weight copies, no batch-1 cross-attention fold). An engine created for > 8 utterances and
                            called with <= 8 still runs the strip step (correct, slower): size engines per batch class */
  int32_t max_ctx;       /* self-attention KV capacity in positions: P + max_length (cf. _get_cache :3254-3309) */
  int32_t max_enc;       /* cross-attention capacity: description (+ prompt if prompt_cross_attention) tokens */
  int32_t max_prompt;    /* prefill capacity in positions per utterance: P + 1 (prompt tokens + the BOS column) */
  int32_t device;        /* HIP device ordinal */
  int32_t num_kv_heads;       /* grouped-query attention, self (repeat_kv :280-289, :449-452): 0 = num_heads (Mini/Large v1) */
  int32_t num_cross_kv_heads; /* cross-attention K/V heads: 0 = num_kv_heads */
  int32_t weights_fp8;        /* 1 (dtype PTTS_BF16 only): the decode step streams OCP e4m3 weight bytes with one power-of-two scale per output
                                 row (ptts_load_weight_fp8) at EVERY batch size: the GEMV step up to 8 utterances, e4m3 MFMA strips above
                                 (converted to bf16 in registers). Only the prefill and the cross-attention q projection inside the fused
                                 cross block read the exact bf16 dequantisation */
  int32_t kv_fp8;             /* (ABI v7) 1 (dtype PTTS_BF16, max_batch > 8 only): the self-attention KV cache holds OCP e4m3 bytes with one power-of-two
                                 scale per (utterance, K/V head, position), quantised when a row is appended. Opt-in numerics mode - the reference has no
                                 quantised cache on this path (modeling_parler_tts.py:3497-3501 raises) - for 64+ utterances per GPU, where the K/V
                                 stream is the bandwidth-bound third of a step */
} ptts_config;

/* Generation parameters: the subset of GenerationConfig that generate() consumes (:3395-3552). */
typedef struct {
  int32_t max_length;      /* total columns incl. the BOS column = 1 + max_new_tokens (:3458-3469) */
  int32_t min_new_tokens;  /* MinNewTokensLengthLogitsProcessor */
  int32_t do_sample;       /* 0 greedy (argmax), 1 multinomial */
  float temperature;       /* TemperatureLogitsWarper (1.0 = off) */
  int32_t top_k;           /* TopKLogitsWarper (0 = off) */
  float top_p;             /* TopPLogitsWarper (1.0 = off) */
  int32_t use_eos_gate;    /* default LogitsProcessorList([ParlerTTSLogitsProcessor]) (:3418, logits_processors.py:6-53) */
  uint64_t seed;           /* Philox key for multinomial sampling */
} ptts_gen_params;

const char* ptts_last_error(void);
int ptts_abi_version(void);

/* ---- decoder LM engine -------------------------------------------------------------------------
 * replaces ParlerTTSForCausalLM (+ the _sample loop around it): modeling_parler_tts.py:1824-1974,
 * :1338-1736, :940-1074, :818-930, logits_processors.py:6-53, :205-276, transformers _sample (:3564). */
int ptts_engine_create(const ptts_config* cfg, ptts_engine** out);
void ptts_engine_destroy(ptts_engine* e);

/* Load one tensor by its reference state-dict name relative to the decoder, e.g.
 * "model.decoder.layers.3.self_attn.q_proj.weight", "model.decoder.embed_tokens.0.weight",
 * "model.decoder.embed_positions.weights", "model.decoder.layer_norm.bias", "lm_heads.4.weight"
 * (or fused "lm_heads.weight" [K*V,H], :1834-1840). `src_dtype` is the dtype of `dev_ptr` (PTTS_F32|PTTS_BF16);
 * the engine converts to its own dtype and re-packs into MFMA fragment order. Replaces
 * load_state_dict via from_pretrained (:2469-2488). */
int ptts_load_weight(ptts_engine* e, const char* name, const void* dev_ptr, int32_t src_dtype,
                     const int64_t* shape, int32_t ndim, void* stream);
/* weights_fp8 engines (BASELINE configs[4]): e4m3 bytes q_dev uint8 [N, K] + scale_dev float32 [N] of one decode-step projection
 * matrix (q/k/v/out_proj, encoder_attn q/out_proj, fc1, fc2, lm_heads.k), with W = q * scale[:, None] exactly equal to the tensor
 * given to ptts_load_weight under the same name (power-of-two scales make the bf16 dequantisation exact). */
int ptts_load_weight_fp8(ptts_engine* e, const char* name, const uint8_t* q_dev, const float* scale_dev, const int64_t* shape,
                         int32_t ndim, void* stream);
/* 0 when every tensor the config requires has been loaded, PTTS_E_MISSING (message lists names) otherwise. */
int ptts_weights_ready(ptts_engine* e);

int ptts_set_gen_params(ptts_engine* e, const ptts_gen_params* gp);

/* Voice prompt (modeling_parler_tts.py:3136-3194: `input_values` -> audio codes -> `decoder_input_ids`; delay pattern over a
 * given prefix :205-276): codes_dev int64 [B*K, T] = the un-delayed codes of T prompt frames. Applies to the NEXT
 * ptts_prefill only, which then also runs the T prefix columns (teacher-forced) before the first sampled column;
 * min_new_tokens / max_length count from the 1 + T given columns as `_sample` does. T = 0 clears a pending prefix. */
int ptts_set_audio_prefix(ptts_engine* e, const int64_t* codes_dev, int32_t B, int32_t T, void* stream);

/* Prefill = the first _sample iteration (:3564; forward :1392-1655 with prompt_hidden_states prepended
 * :1437-1439): computes cross K/V from the (already projected + masked, :3086-3093) encoder states, runs
 * P+1 positions (prompt embeddings + the BOS column) through the stack, fills the self KV cache, leaves
 * the step-0 logits in the engine. If `sample` != 0 it also runs the device-side tail (processors +
 * select + append), i.e. the first token is materialised when this call's work completes (TTFT).
 *   enc_dev        [B, N, H] float32   encoder_hidden_states
 *   enc_mask_dev   [B, N] int32 or NULL (attention_mask; 1 = keep)
 *   prompt_dev     [B, P, H] float32 or NULL (embed_prompts(prompt_input_ids), :3100)
 *   prompt_mask_dev[B, P] int32 or NULL (prompt_attention_mask) */
int ptts_prefill(ptts_engine* e, const float* enc_dev, const int32_t* enc_mask_dev, const float* prompt_dev,
                 const int32_t* prompt_mask_dev, int32_t B, int32_t N, int32_t P, int32_t sample, void* stream);

/* Blocks the calling thread until the first token of the last ptts_prefill(sample != 0) is materialised on the device (an event recorded
 * right after the sampler tail): time-to-first-token as SURVEY.md §8(d) defines it. Work enqueued behind the tail by the same call (the
 * static cross-attention fold for the decode steps) is NOT waited for. Replaces the host sync a caller of the reference would place after
 * the first `_sample` iteration (:3564). */
int ptts_first_token_sync(ptts_engine* e);
/* (ABI v7) Measurement hook for the time-to-first-token breakdown (bench.py `ttft`), last ptts_prefill(sample != 0): GPU time in ms from the
 * moment the stream reached the call's first piece of work to the start of the sampler tail (prefill_ms: staging copies + the P + 1-position
 * forward, as it ran IN SEQUENCE behind whatever the caller enqueued before), and of the tail itself up to the first-token event (tail_ms).
 * Synchronises on that event. */
int ptts_first_token_times(ptts_engine* e, float* prefill_ms, float* tail_ms);

/* n_steps iterations of {embed(delay-masked last column) -> layers -> LM heads -> tail}; replayed from a
 * captured hipGraph. Steps after every row has finished are no-ops (device-side check), so callers may
 * over-run and poll ptts_state() every few dozen steps instead of syncing per token (:_sample's
 * `unfinished_sequences.max() == 0` host sync). One graph per (batch, 64-position context bucket) is captured on first
 * use and kept for the engine's life: the only kernel argument that changes is the attention fetch bound, an upper bound of
 * the context the host knows without reading device state. */
int ptts_decode_steps(ptts_engine* e, int32_t n_steps, void* stream);

/* Host-visible loop state (SYNCHRONISES the stream): number of columns written so far (incl. BOS),
 * and whether every row has finished (EOS or max_length). */
int ptts_state(ptts_engine* e, int32_t* cur_len, int32_t* all_finished, void* stream);

/* Raw (un-masked) token ids as _sample stores them: int64 [B*K, row_stride] device buffer owned by the
 * engine, column 0 = BOS. Valid columns: [0, cur_len). */
int ptts_ids(ptts_engine* e, int64_t** ids_dev, int32_t* row_stride);

/* ---- hooks for user LogitsProcessorList / StoppingCriteria (the `logits_processor=` argument, :3418) ----
 * ptts_step_forward: forward only (no tail) for the next position; logits fp32 [B*K, V] at *logits_dev.
 * ptts_push_tokens : append caller-chosen raw tokens [B*K] int64 (already pad-substituted) and per-row
 *                    finished flags [B*K] int32 (1 = this row is now finished). */
int ptts_step_forward(ptts_engine* e, void* stream);
int ptts_logits(ptts_engine* e, float** logits_dev);
int ptts_push_tokens(ptts_engine* e, const int64_t* tokens_dev, const int32_t* finished_dev, void* stream);

/* (ABI v7) Kernel nodes per decode step of the step graph captured last (0 before the first capture): what the `latency_model` of bench.py counts,
 * and how the tests check that a fused node really is in the step. */
int ptts_debug_graph_nodes(ptts_engine* e, int32_t* nodes);
/* Debug / parity probes: residual stream after the last forward, fp32 [rows, H]. */
int ptts_debug_hidden(ptts_engine* e, float** hidden_dev, int32_t* rows);

/* ---- DAC decode engine ----------------------------------------------------------------------------
 * replaces DACModel.decode (dac_wrapper/modeling_dac.py:106-142): quantizer.from_codes (:138) + the
 * descript-audio-codec decoder stack (:139). */
typedef struct {
  int32_t num_codebooks, codebook_size, codebook_dim, latent_dim, decoder_dim;
  int32_t num_rates;
  int32_t rates[8];
  int32_t compute_dtype; /* PTTS_F32: exact-f32 MFMA and exact sinf (parity, RMS <= 1e-4); PTTS_BF16: bf16 MFMA operands and bf16
                            activations between layers, fp32 accumulate / bias / residual stream, Snake's sin on v_sin_f32. Contract of the
                            bf16 mode: every decoder stage within 2e-4 relative RMS of the bf16-operand oracle on IDENTICAL inputs (measured
                            3-5e-5, tests/test_dac_stage_parity_gpu.py) and the whole decode within 2e-2 of that oracle end to end (measured
                            9.8e-3 at 860 frames: the floor of any bf16 evaluation of this stack, profiles/r04_dac_bf16_sensitivity.txt) */
  int32_t max_batch, max_frames;
  int32_t device;
  int32_t encoder_dim;   /* 0: decode only; > 0: also build the encoder (descript default 64) for ptts_dac_encode */
} ptts_dac_config;

int ptts_dac_create(const ptts_dac_config* cfg, ptts_dac** out);
void ptts_dac_destroy(ptts_dac* d);
/* names relative to dac.model.DAC, weight-norm ALREADY folded by the caller (w = g*v/||v||,
 * modeling_dac.py:148-164): "quantizer.quantizers.i.codebook.weight", "quantizer.quantizers.i.out_proj.{weight,bias}",
 * "decoder.model.N...{weight,bias,alpha}". float32 device tensors. */
int ptts_dac_load_weight(ptts_dac* d, const char* name, const float* dev_ptr, const int64_t* shape, int32_t ndim, void* stream);
int ptts_dac_weights_ready(ptts_dac* d);
/* codes_dev int64 [B, K, T] -> wave_dev float32 [B, hop*T]  (hop = prod(rates)). */
int ptts_dac_decode(ptts_dac* d, const int64_t* codes_dev, float* wave_dev, int32_t B, int32_t T, void* stream);
/* (ABI v6) The per-sample branch of generate() (modeling_parler_tts.py:3615-3647: after the special-id filter every utterance has its own
 * number of frames; the reference decodes them one at a time and zero-pads with pad_sequence) as two enqueued calls, no host round trip
 * between them:
 *   ptts_dac_compact_codes  - per utterance, drops every frame in which any codebook holds an id outside [0, codebook_size) (:3627-3636)
 *                             and keeps the rest in order: codes_in/out int64 [B][K][T] (must not alias), frames_out int32 [B] ON THE DEVICE;
 *   ptts_dac_decode_ragged  - ONE decode pass of codes [B][K][T] in which utterance b is decoded as exactly frames_dev[b] (<= T) frames:
 *                             rows past its length are the convolutions' zero padding, tiles past it exit at once; wave_dev float32
 *                             [B][hop*T], zero beyond hop*frames_dev[b] (what pad_sequence(..., padding_value=0) yields, :3643-3647). */
int ptts_dac_compact_codes(ptts_dac* d, const int64_t* codes_in_dev, int64_t* codes_out_dev, int32_t* frames_out_dev, int32_t B, int32_t T, void* stream);
int ptts_dac_decode_ragged(ptts_dac* d, const int64_t* codes_dev, const int32_t* frames_dev, float* wave_dev, int32_t B, int32_t T, void* stream);
/* Streaming / chunked decode (parler_tts/streamer.py:66-131: `apply_delay_pattern_mask` there re-decodes the whole token
 * cache at every `play_steps`; SURVEY.md §8(b) `ptts_dac_decode_chunk`): the window of frames [first_frame - halo (clamped at 0),
 * first_frame + n_frames) of codes_dev int64 [B, K, codes_ld] is decoded and the samples of its frames [first_frame,
 * first_frame + n_emit) are written to wave_dev float32 rows of stride wave_ld samples (n_emit <= 0: n_frames; wave_ld <= 0:
 * hop * n_emit). With halo >= the decoder's one-sided receptive field (13 frames for strides 8, 8, 4, 2):
 *   - n_emit == n_frames: exactly the samples a decode of frames [0, first_frame + n_frames) has there (the streamer's case:
 *     nothing exists yet to the right);
 *   - n_emit == n_frames - halo: exactly the samples of the FULL utterance decode (the frames kept back are the right halo),
 *     which lets generate() decode finished frames chunk by chunk on a second stream while ptts_decode_steps keeps running. */
int ptts_dac_decode_chunk(ptts_dac* d, const int64_t* codes_dev, int64_t codes_ld, int32_t first_frame, int32_t n_frames,
                          int32_t halo, float* wave_dev, int64_t wave_ld, int32_t n_emit, int32_t B, void* stream);
/* DACModel.encode for voice prompts (dac_wrapper/modeling_dac.py:33-104, used by modeling_parler_tts.py:3136-3194):
 * wave_dev float32 [B, L], L a multiple of the hop (the caller applies model.preprocess's right zero padding, :64)
 * -> codes_dev int64 [B, n_quantizers, L/hop]  (model.encode :95: encoder stack + residual VQ nearest-neighbour search).
 * n_quantizers <= 0 means all. Extra weights: "encoder.block...", "quantizer.quantizers.i.in_proj.{weight,bias}". */
int ptts_dac_encode(ptts_dac* d, const float* wave_dev, int64_t* codes_dev, int32_t B, int32_t L, int32_t n_quantizers, void* stream);
/* Debug / parity probe: latents z of the last encode, channels-last fp32 [B, L/hop, latent_dim]. */
int ptts_dac_debug_latents(ptts_dac* d, float** latents_dev);
/* (ABI v6) Parity probe: ptts_dac_decode stopped after `stage` (0 = decoder.model.0, then per up-sampling block its transposed conv and its
 * three residual units: 1 + 4 * num_rates stages). Returns the engine's own buffers with that stage's outputs, valid until the next call:
 * act [B][rows][channels] (bf16 bits if *act_is_bf16 else fp32: the Snake'd activation the next conv reads), raw [B][rows][channels] fp32
 * (the residual stream; null for stage 0). Replaces nothing in the reference: it exists so that each conv kernel can be compared with the
 * oracle's restatement of ONE layer (dac_wrapper/modeling_dac.py:139 -> descript DecoderBlock / ResidualUnit) on identical inputs. */
int ptts_dac_debug_decode_upto(ptts_dac* d, const int64_t* codes_dev, int32_t B, int32_t T, int32_t stage, void* stream, void** act_dev,
                               int32_t* act_is_bf16, float** raw_dev, int32_t* rows, int32_t* channels);

/* ---- T5 description encoder (ABI v7) --------------------------------------------------------------------
 * replaces the `self.text_encoder(input_ids=..., attention_mask=...)` call of generate() (modeling_parler_tts.py:3048-3097; the module is
 * transformers' T5EncoderModel, built at :2345-2348 - google/flan-t5-large for Mini-v1 / Large-v1, training/README.md:91) on the
 * time-to-first-token path: T5Stack.forward = shared embedding -> num_layers x [T5LayerSelfAttention, T5LayerFF (gated gelu_new)] ->
 * final T5LayerNorm, with the bucketed relative-position bias of block 0 shared by every block. SURVEY.md section 8(f) rank 2. */
typedef struct {
  int32_t vocab_size, d_model, d_kv, d_ff, num_layers, num_heads; /* T5Config; d_kv must be 64, feed_forward_proj "gated-gelu" */
  int32_t rel_buckets, rel_max_distance;                          /* relative_attention_num_buckets (32), relative_attention_max_distance (128) */
  float layer_norm_eps;                                           /* layer_norm_epsilon (1e-6) */
  int32_t dtype;     /* PTTS_F32 (parity mode) | PTTS_BF16 (bf16 weights and GEMM operands, fp32 accumulate / residual stream / attention) */
  int32_t max_batch; /* descriptions per call */
  int32_t max_len;   /* tokens per description */
  int32_t device;
} ptts_t5_config;

int ptts_t5_create(const ptts_t5_config* cfg, ptts_t5** out);
void ptts_t5_destroy(ptts_t5* e);
/* One tensor by its transformers T5EncoderModel state-dict name: "shared.weight" (= "encoder.embed_tokens.weight"),
 * "encoder.block.L.layer.0.SelfAttention.{q,k,v,o}.weight", "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
 * "encoder.block.L.layer.0.layer_norm.weight", "encoder.block.L.layer.1.DenseReluDense.{wi_0,wi_1,wo}.weight",
 * "encoder.block.L.layer.1.layer_norm.weight", "encoder.final_layer_norm.weight". Replaces load_state_dict via from_pretrained (:2469-2488). */
int ptts_t5_load_weight(ptts_t5* e, const char* name, const void* dev_ptr, int32_t src_dtype, const int64_t* shape, int32_t ndim, void* stream);
int ptts_t5_weights_ready(ptts_t5* e);
/* ids_dev int64 [B, N], mask_dev int32 [B, N] or NULL (attention_mask; 1 = keep) -> out_dev float32 [B, N, d_model] = last_hidden_state with
 * the masked positions zeroed (:3093-3097). Replayed from one captured hipGraph per (B, N, masked?). */
int ptts_t5_encode(ptts_t5* e, const int64_t* ids_dev, const int32_t* mask_dev, int32_t B, int32_t N, float* out_dev, void* stream);
/* Host-only (no device): T5Attention._relative_position_bucket for the bidirectional encoder, the function the bias table is built from. */
int32_t ptts_t5_relative_bucket(int32_t relative_position, int32_t num_buckets, int32_t max_distance);
/* (ABI v8) Kernel nodes of the encoder graph captured last by ptts_t5_encode (0 before the first capture, or with graphs off): what bench.py's
 * `ttft.launches` reports instead of a formula. */
int ptts_t5_debug_graph_nodes(ptts_t5* e, int32_t* nodes);

#ifdef __cplusplus
}
#endif
#endif /* PTTS_H_ */
