// GEMV step instances: fp32 parity engine (one utterance) + the mode dispatcher.
#define GV_WT float
#define GV_W8 false
#define GV_FN ptts_gemv_launch_f32
#define GV_QA_FN ptts_qkvattn_launch_f32
#define GV_XQ_FN ptts_xqattn_launch_f32
#define GV_XA_FN ptts_xfoldattn_launch_f32
#include "ptts_gemv_launch.inc"

int ptts_gemv_launch_bf16(int pro, int epi, int S, GemvArgs a, hipStream_t st);
int ptts_gemv_launch_w8(int pro, int epi, int S, GemvArgs a, hipStream_t st);

int ptts_qkvattn_launch_bf16(QkvAttnArgs a, hipStream_t st);
int ptts_qkvattn_launch_w8(QkvAttnArgs a, hipStream_t st);

int ptts_xqattn_launch_bf16(XqAttnArgs a, hipStream_t st);
int ptts_xqattn_launch_w8(XqAttnArgs a, hipStream_t st);
int ptts_xqattn_launch(int mode, XqAttnArgs a, hipStream_t st) {
  if (mode == GV_BF16) return ptts_xqattn_launch_bf16(a, st);
  if (mode == GV_BF16_W8) return ptts_xqattn_launch_w8(a, st);
  return ptts_xqattn_launch_f32(a, st);
}
bool ptts_xqattn_ok(int H, int mode) { return ptts_qkvattn_ok(H, mode); }

int ptts_qkvattn_launch(int mode, QkvAttnArgs a, hipStream_t st) {
  if (mode == GV_BF16) return ptts_qkvattn_launch_bf16(a, st);
  if (mode == GV_BF16_W8) return ptts_qkvattn_launch_w8(a, st);
  return ptts_qkvattn_launch_f32(a, st);
}
bool ptts_qkvattn_ok(int H, int mode) {
  if (mode == GV_F32) return H == 512 || H == 1024;
  return H == 512 || H == 1024 || H == 1536;
}
int ptts_qkvattn_rows_per_split(int mode) { return 8 * 4 * (mode == GV_F32 ? 4 : 8); }

int ptts_xfoldattn_launch_bf16(XfoldAttnArgs a, hipStream_t st);
int ptts_xfoldattn_launch(int mode, XfoldAttnArgs a, hipStream_t st) {
  if (mode == GV_BF16) return ptts_xfoldattn_launch_bf16(a, st);
  if (mode == GV_F32) return ptts_xfoldattn_launch_f32(a, st);
  return -1;  // the folded matrices are never e4m3
}
bool ptts_xfoldattn_ok(int H, int nheads, int mode) {
  if (nheads > GV_PMAX) return false;
  if (mode == GV_F32) return H == 512;
  return H == 512 || H == 1024 || H == 1536;
}

int ptts_gemv_launch(int mode, int pro, int epi, int S, GemvArgs a, hipStream_t st) {
  if (mode == GV_BF16) return ptts_gemv_launch_bf16(pro, epi, S, a, st);
  if (mode == GV_BF16_W8) return ptts_gemv_launch_w8(pro, epi, S, a, st);
  return ptts_gemv_launch_f32(pro, epi, S, a, st);
}

bool ptts_gemv_k_ok(int K, int mode) {
  const int epl = mode == GV_F32 ? 4 : 8;
  if (K % (64 * epl)) return false;
  const int nch = K / (64 * epl);
  return nch == 1 || nch == 2 || nch == 3 || nch == 4 || nch == 6 || nch == 8 || nch == 12 || nch == 16 || nch == 24;
}
