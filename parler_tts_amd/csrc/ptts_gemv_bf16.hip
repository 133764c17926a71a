// GEMV step instances: bf16 engine (bf16 weights, bf16 activations into the dot products), batch 1 and 2..4.
#define GV_WT bf16_t
#define GV_W8 false
#define GV_MULTI 1
#define GV_FN ptts_gemv_launch_bf16
#define GV_QA_FN ptts_qkvattn_launch_bf16
#define GV_XQ_FN ptts_xqattn_launch_bf16
#define GV_XA_FN ptts_xfoldattn_launch_bf16
#include "ptts_gemv_launch.inc"
