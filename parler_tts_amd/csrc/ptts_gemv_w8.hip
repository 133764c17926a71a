// GEMV step instances: bf16 engine with OCP e4m3 weights (one power-of-two scale per output row), batch 1 and 2..4.
#define GV_WT bf16_t
#define GV_W8 true
#define GV_MULTI 1
#define GV_FN ptts_gemv_launch_w8
#define GV_QA_FN ptts_qkvattn_launch_w8
#define GV_XQ_FN ptts_xqattn_launch_w8
#include "ptts_gemv_launch.inc"
