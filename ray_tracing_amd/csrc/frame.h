// frame.h -- RtRenderParams -> RtFrameDev: the per-call constants the kernels read, including the pixel set of the call
// (RtRenderParams.tile / .stripe).  Shared by render.hip and the test-only host emulation, so that both number the pixels of a
// shard the same way.
#pragma once
#include <string.h>
#include "../../include/rt_hip.h"
#include "rt_types.h"

// Rows that a round-robin deal of bands of `band` rows gives to shard `shard` of `n_shards` out of `rows` rows.
static inline int64_t rt_stripe_rows(int64_t rows, int64_t band, int64_t n_shards, int64_t shard) {
    if (band <= 0) return rows;
    const int64_t n_bands = (rows + band - 1) / band;                    // the last band may be short
    if (shard >= n_bands) return 0;
    const int64_t mine = (n_bands - 1 - shard) / n_shards + 1;           // bands shard, shard + n, ...
    const bool has_last = (n_bands - 1) % n_shards == shard;
    return mine * band - (has_last ? n_bands * band - rows : 0);
}
// 0, or the reason the pixel selection of `p` is invalid.
static inline const char* rt_frame_check(const RtRenderParams* p) {
    const int32_t* t = p->tile; const int32_t* s = p->stripe;
    const bool no_tile = t[0] == 0 && t[1] == 0 && t[2] == 0 && t[3] == 0;
    if (!no_tile && (t[0] < 0 || t[1] < 0 || t[2] < 1 || t[3] < 1 || (int64_t)t[0] + t[2] > p->width || (int64_t)t[1] + t[3] > p->height))
        return "tile = (x0, y0, w, h) must lie inside the image (all zero = the whole image)";
    const bool no_stripe = s[0] == 0 && s[1] == 0 && s[2] == 0;
    if (!no_stripe && (s[0] < 1 || s[1] < 1 || s[2] < 0 || s[2] >= s[1]))
        return "stripe = (band_rows >= 1, n_shards >= 1, 0 <= shard < n_shards) (all zero = every row)";
    return 0;
}
static inline void rt_make_frame(const RtRenderParams* p, RtFrameDev& fr) {
    memset(&fr, 0, sizeof(fr));
    fr.width = p->width; fr.height = p->height; fr.n_pixels = p->width * p->height;
    fr.max_depth = p->max_depth;
    fr.key0 = (uint32_t)p->seed; fr.key1 = (uint32_t)(p->seed >> 32);
    fr.inv_wm1 = (float)(1.0 / (double)(p->width - 1)); fr.inv_hm1 = (float)(1.0 / (double)(p->height - 1));
    for (int c = 0; c < 3; ++c) fr.background[c] = p->background[c];
    const int32_t* t = p->tile; const int32_t* s = p->stripe;
    const bool no_tile = t[0] == 0 && t[1] == 0 && t[2] == 0 && t[3] == 0;
    const bool no_stripe = s[0] == 0 && s[1] == 0 && s[2] == 0;
    fr.n_set = fr.n_pixels; fr.subset = 0;
    fr.set_x0 = 0; fr.set_y0 = 0; fr.set_w = p->width; fr.band = 0; fr.n_shards = 1; fr.shard = 0;
    if (!no_tile || !no_stripe) {
        fr.subset = 1;
        int rows = p->height;
        if (!no_tile) { fr.set_x0 = t[0]; fr.set_y0 = t[1]; fr.set_w = t[2]; rows = t[3]; }
        if (!no_stripe) { fr.band = s[0]; fr.n_shards = s[1]; fr.shard = s[2]; }
        fr.n_set = (int32_t)(rt_stripe_rows(rows, fr.band, fr.n_shards, fr.shard) * fr.set_w);
    }
    fr.inv_n_set = fr.n_set > 0 ? 1.0 / (double)fr.n_set : 0.0;
    fr.inv_set_w = 1.0 / (double)fr.set_w;
}
