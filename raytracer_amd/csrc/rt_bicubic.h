// rt_bicubic.h -- the encoder's bicubic upscale (reference scene.c:188-202 around utils/image.c:4-78),
// one column strip at a time, written once for the device kernel and for the host build the CPU tests
// use.  Compile with FMA contraction off: every operator below is one IEEE rounding, in the
// reference's order.
//
// The reference interpolates every output pixel from scratch: four horizontal cubics per channel
// (one per source row of the 4x4 patch), then one vertical cubic per channel.  The horizontal result
// of a source row depends only on the output COLUMN (x1, fx), so consecutive output rows of one
// column share it: a strip walks down its column keeping the four most recent row results and
// computes only the rows that enter the window.  The values are the reference's, bit for bit; only
// the repetition is gone.
#pragma once
#include "rt_exact.h"

#ifndef RT_UNROLL
#if defined(__HIPCC__) || defined(__clang__)
#define RT_UNROLL _Pragma("unroll")
#else
#define RT_UNROLL _Pragma("GCC unroll 8")
#endif
#endif

namespace rt {

// image.c:4-11.  Division by 2.0f is exact, so its placement does not matter; the additions are
// left to right as written there.
RT_HD float cubic_hermite(float A, float B, float C, float D, float t) {
    const float a = -A / 2.0f + (3.0f * B) / 2.0f - (3.0f * C) / 2.0f + D / 2.0f;
    const float b = A - (5.0f * B) / 2.0f + 2.0f * C - D / 2.0f;
    const float c = -A / 2.0f + C / 2.0f;
    return a * t * t * t + b * t * t + c * t + B;
}

struct Rgb3 { float r, g, b; };

// Source coordinate of output index i (scene.c:191-192) split into integer and fractional part
// (image.c:26-29).
RT_HD void source_coord(int i, int n_src, int n_dst, int& whole, float& frac) {
    const float s = (float)i * ((float)n_src - 1.0f) / ((float)n_dst - 1.0f);
    whole = (int)s;
    frac = s - (float)whole;
}

RT_HD int clamp_index(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }     // image.c:15-18

// Horizontal pass of one source row for one output column (image.c:43-65): `row` points at the
// row's first byte, off[k] is the byte offset of the (clamped) pixel x1-1+k.
RT_HD Rgb3 row_pass(const uint8_t* row, const int off[4], float fx) {
    const uint8_t *p0 = row + off[0], *p1 = row + off[1], *p2 = row + off[2], *p3 = row + off[3];
    Rgb3 o;
    o.r = cubic_hermite((float)p0[0], (float)p1[0], (float)p2[0], (float)p3[0], fx);
    o.g = cubic_hermite((float)p0[1], (float)p1[1], (float)p2[1], (float)p3[1], fx);
    o.b = cubic_hermite((float)p0[2], (float)p1[2], (float)p2[2], (float)p3[2], fx);
    return o;
}

RT_HD uint32_t quantise(float v) {                                                    // image.c:68-75
    int q = (int)(v + 0.5f);
    q = q < 0 ? 0 : (q > 255 ? 255 : q);
    return (uint32_t)q;
}

// Vertical pass and packing (image.c:68-77).
RT_HD uint32_t column_pass(const Rgb3 w[4], float fy) {
    const uint32_t r = quantise(cubic_hermite(w[0].r, w[1].r, w[2].r, w[3].r, fy));
    const uint32_t g = quantise(cubic_hermite(w[0].g, w[1].g, w[2].g, w[3].g, fy));
    const uint32_t b = quantise(cubic_hermite(w[0].b, w[1].b, w[2].b, w[3].b, fy));
    return 0xFF000000u | (r << 16) | (g << 8) | b;
}

// Output pixels (x, y0 .. y0+rows-1) of one frame.  `store(y, argb)` receives the results.
// The window advance is decided from y alone, so in a kernel whose lanes differ only in x every
// branch here is uniform over the wave.
template <class Store>
RT_HD void upscale_strip(const uint8_t* frame, int width, int height, int scaled_width, int scaled_height,
                         int x, int y0, int rows, Store store) {
    int x1; float fx;
    source_coord(x, width, scaled_width, x1, fx);
    int off[4];
    RT_UNROLL
    for (int k = 0; k < 4; k++) off[k] = clamp_index(x1 - 1 + k, width) * 3;
    const size_t pitch = (size_t)width * 3;
    Rgb3 w[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int have = 0;
    bool primed = false;
    for (int y = y0; y < y0 + rows && y < scaled_height; y++) {
        int y1; float fy;
        source_coord(y, height, scaled_height, y1, fy);
        const int step = primed ? y1 - have : 4;
        // slot k holds source row y1-1+k (clamped); rows still inside the window move up
        if (step == 1) { w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; }
        else if (step == 2) { w[0] = w[2]; w[1] = w[3]; }
        else if (step == 3) { w[0] = w[3]; }
        if (step != 0) {
            const int keep = (step < 0 || step > 3) ? 0 : 4 - step;
            RT_UNROLL
            for (int k = 0; k < 4; k++)
                if (k >= keep) w[k] = row_pass(frame + (size_t)clamp_index(y1 - 1 + k, height) * pitch, off, fx);
        }
        have = y1; primed = true;
        store(y, column_pass(w, fy));
    }
}

}  // namespace rt
