// Helpers shared by the entry-point translation units (api*.hip).
#pragma once
#include "common.hpp"

// stage timers are cleared per pipeline: the KLT entry points own [ST_MINMAX, ST_LK], ZNCC owns ST_ZNCC
enum { RESET_NONE = 0, RESET_KLT = 1, RESET_ZNCC = 2 };
int begin_call(km_ctx *c, int reset = RESET_NONE);   // start of every entry point: device, pending uploads, stale jobs, stage timers
int check_image(km_ctx *c, const void *p, int H, int W, ptrdiff_t stride, const char *what);
int check_params(km_ctx *c, const km_klt_params *p);
int frame_block_free(km_ctx *c);                     // WS_FRAME may be rewritten once the previous submitted frame's block has left
