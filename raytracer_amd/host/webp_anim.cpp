// webp_anim.cpp -- animated-WebP assembly for save_scene (reference scene.c:164-224), pipelined.
//
// The reference hands every upscaled frame to libwebpmux's WebPAnimEncoder, one after the other on one
// thread.  libwebpmux and the WebP headers are not part of this image; libwebp.so.7 is, and its simple
// one-call encoders (WebPEncodeRGB / WebPEncodeLosslessRGB: plain pointers and numbers, no structs)
// return a complete single-image file whose bitstream chunk can be lifted out.  An animation is then a
// RIFF container around those chunks (VP8X header with the animation bit, ANIM, one ANMF per frame),
// which is written here directly.  Every frame is a full-canvas key frame, so frames are independent:
// they are encoded on several host threads while the caller is still producing (upscaling, rendering,
// receiving) later frames, and appended to the file in order.
//
// Without the encoder symbols the writer falls back to a stream of binary PPM images (P6, one per
// frame, concatenated) under the same name with ".ppm" appended.
#include "raytracer_host.h"
#include "../../include/rt_mi355.h"

#include <dlfcn.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef size_t (*encode_lossy_fn)(const uint8_t*, int, int, int, float, uint8_t**);
typedef size_t (*encode_lossless_fn)(const uint8_t*, int, int, int, uint8_t**);
typedef void (*free_fn)(void*);

struct Encoder {
    encode_lossy_fn lossy = nullptr;
    encode_lossless_fn lossless = nullptr;
    free_fn release = nullptr;
    bool ready() const { return lossy && lossless && release; }
};

const Encoder& encoder() {
    static Encoder e;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* off = getenv("RT_HOST_NO_WEBP");             // for exercising the PPM fallback
        if (off && off[0] == '1') return;
        static const char* names[] = {"libwebp.so.7", "libwebp.so", "libwebp.so.6", nullptr};
        for (int i = 0; names[i] && !e.ready(); i++) {
            void* h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            e.lossy = (encode_lossy_fn)dlsym(h, "WebPEncodeRGB");
            e.lossless = (encode_lossless_fn)dlsym(h, "WebPEncodeLosslessRGB");
            e.release = (free_fn)dlsym(h, "WebPFree");
        }
    });
    return e;
}

void put24(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x & 0xFF); v.push_back((x >> 8) & 0xFF); v.push_back((x >> 16) & 0xFF); }
void put32(std::vector<uint8_t>& v, uint32_t x) { put24(v, x); v.push_back((x >> 24) & 0xFF); }
void put_tag(std::vector<uint8_t>& v, const char* t) { v.insert(v.end(), t, t + 4); }
uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct Job {
    int index;
    int duration_ms;
    std::vector<uint8_t> rgb;
};

}  // namespace

struct rt_host_animation {
    FILE* file = nullptr;
    bool webp = false;                 // false: PPM stream
    int width = 0, height = 0;
    float quality = 75.0f;             // < 0: lossless
    int status = 0;
    int added = 0;

    std::mutex lock;
    std::condition_variable work_ready, room, flushed;
    std::deque<Job> queue;
    std::map<int, std::vector<uint8_t>> done;      // frame index -> bytes to append
    int next_to_write = 0;
    int in_flight = 0;
    int max_in_flight = 0;
    bool closing = false;
    std::vector<std::thread> workers;

    // One frame as an ANMF chunk: position 0,0, full canvas, no blending, no disposal.
    bool encode(const Job& j, std::vector<uint8_t>& out) {
        if (!webp) {
            char head[64];
            const int n = snprintf(head, sizeof(head), "P6\n%d %d\n255\n", width, height);
            out.assign(head, head + n);
            out.insert(out.end(), j.rgb.begin(), j.rgb.end());
            return true;
        }
        const Encoder& e = encoder();
        uint8_t* bytes = nullptr;
        const size_t n = quality < 0.0f ? e.lossless(j.rgb.data(), width, height, width * 3, &bytes)
                                        : e.lossy(j.rgb.data(), width, height, width * 3, quality, &bytes);
        if (n < 20 || !bytes || memcmp(bytes, "RIFF", 4) != 0 || memcmp(bytes + 8, "WEBP", 4) != 0) {
            if (bytes) e.release(bytes);
            return false;
        }
        // keep the bitstream chunks of the single-image file (an RGB source has no alpha, but take ALPH if present)
        std::vector<uint8_t> stream;
        for (size_t off = 12; off + 8 <= n;) {
            const uint32_t size = get32(bytes + off + 4);
            const size_t padded = (size_t)size + (size & 1u);
            if (off + 8 + size > n) break;
            if (!memcmp(bytes + off, "VP8 ", 4) || !memcmp(bytes + off, "VP8L", 4) || !memcmp(bytes + off, "ALPH", 4)) {
                stream.insert(stream.end(), bytes + off, bytes + off + 8 + size);
                if (size & 1u) stream.push_back(0);
            }
            off += 8 + padded;
        }
        e.release(bytes);
        if (stream.empty()) return false;
        out.clear();
        put_tag(out, "ANMF");
        put32(out, (uint32_t)(16 + stream.size()));
        put24(out, 0); put24(out, 0);
        put24(out, (uint32_t)(width - 1)); put24(out, (uint32_t)(height - 1));
        put24(out, (uint32_t)(j.duration_ms < 0 ? 0 : (j.duration_ms > 0xFFFFFF ? 0xFFFFFF : j.duration_ms)));
        out.push_back(0x02);
        out.insert(out.end(), stream.begin(), stream.end());
        return true;
    }

    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(lock);
                work_ready.wait(g, [&] { return closing || !queue.empty(); });
                if (queue.empty()) return;
                j = std::move(queue.front());
                queue.pop_front();
            }
            std::vector<uint8_t> bytes;
            const bool ok = encode(j, bytes);
            std::unique_lock<std::mutex> g(lock);
            if (!ok) status = RT_ERR_INVALID;
            done[j.index] = std::move(bytes);
            // whoever finishes the next frame in file order appends everything that is ready
            while (!done.empty() && done.begin()->first == next_to_write) {
                const std::vector<uint8_t>& b = done.begin()->second;
                if (!b.empty() && fwrite(b.data(), 1, b.size(), file) != b.size()) status = RT_ERR_INVALID;
                done.erase(done.begin());
                next_to_write++;
                in_flight--;
            }
            room.notify_all();
            flushed.notify_all();
        }
    }
};

extern "C" {

int rt_host_animation_has_webp(void) { return encoder().ready() ? 1 : 0; }

int rt_host_animation_open(const char* filename, int width, int height, float quality, int threads, rt_host_animation** out) {
    if (!filename || !out || width <= 0 || height <= 0 || width > 16383 || height > 16383) return RT_ERR_INVALID;
    *out = nullptr;
    rt_host_animation* a = new rt_host_animation;
    a->width = width; a->height = height; a->quality = quality;
    a->webp = encoder().ready();
    std::string path = filename;
    if (!a->webp) {
        path += ".ppm";
        fprintf(stderr, "save_scene: no WebP encoder in this process (libwebp.so not found); writing PPM frames to %s\n", path.c_str());
    }
    a->file = fopen(path.c_str(), "wb");
    if (!a->file) { delete a; return RT_ERR_INVALID; }                       // like the reference: nothing is written (scene.c:215-216)
    if (a->webp) {
        std::vector<uint8_t> h;
        put_tag(h, "RIFF"); put32(h, 0); put_tag(h, "WEBP");                   // size patched by close
        put_tag(h, "VP8X"); put32(h, 10); h.push_back(0x02); put24(h, 0);      // animation bit
        put24(h, (uint32_t)(width - 1)); put24(h, (uint32_t)(height - 1));
        put_tag(h, "ANIM"); put32(h, 6); put32(h, 0xFFFFFFFFu); h.push_back(0); h.push_back(0);   // white background, loop forever: WebPAnimEncoderOptionsInit's defaults (scene.c:169)
        if (fwrite(h.data(), 1, h.size(), a->file) != h.size()) { fclose(a->file); delete a; return RT_ERR_INVALID; }
    }
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        if (threads > 16) threads = 16;
        if (threads < 1) threads = 1;
    }
    a->max_in_flight = 2 * threads + 2;
    for (int i = 0; i < threads; i++) a->workers.emplace_back([a] { a->run(); });
    *out = a;
    return RT_OK;
}

int rt_host_animation_add(rt_host_animation* a, const void* pixels, int format, int duration_ms) {
    if (!a || !pixels || (format != 0 && format != 1)) return RT_ERR_INVALID;
    Job j;
    j.duration_ms = duration_ms;
    const size_t n = (size_t)a->width * (size_t)a->height;
    j.rgb.resize(n * 3);
    if (format == 0) memcpy(j.rgb.data(), pixels, n * 3);
    else {
        const uint32_t* argb = (const uint32_t*)pixels;                         // WebPPicture.argb words (scene.c:194)
        for (size_t i = 0; i < n; i++) {
            const uint32_t p = argb[i];
            j.rgb[3 * i] = (uint8_t)(p >> 16); j.rgb[3 * i + 1] = (uint8_t)(p >> 8); j.rgb[3 * i + 2] = (uint8_t)p;
        }
    }
    std::unique_lock<std::mutex> g(a->lock);
    a->room.wait(g, [&] { return a->in_flight < a->max_in_flight; });
    j.index = a->added++;
    a->in_flight++;
    a->queue.push_back(std::move(j));
    a->work_ready.notify_one();
    return a->status;
}

int rt_host_animation_close(rt_host_animation* a) {
    if (!a) return RT_ERR_INVALID;
    {
        std::unique_lock<std::mutex> g(a->lock);
        a->flushed.wait(g, [&] { return a->next_to_write == a->added; });
        a->closing = true;
        a->work_ready.notify_all();
    }
    for (auto& t : a->workers) t.join();
    int rc = a->status;
    if (a->webp) {
        const long end = ftell(a->file);
        uint8_t size[4];
        const uint32_t riff = (uint32_t)(end - 8);
        size[0] = riff & 0xFF; size[1] = (riff >> 8) & 0xFF; size[2] = (riff >> 16) & 0xFF; size[3] = (riff >> 24) & 0xFF;
        if (end < 0 || fseek(a->file, 4, SEEK_SET) != 0 || fwrite(size, 1, 4, a->file) != 4) rc = RT_ERR_INVALID;
    }
    if (fclose(a->file) != 0) rc = RT_ERR_INVALID;
    delete a;
    return rc;
}

}  // extern "C"
