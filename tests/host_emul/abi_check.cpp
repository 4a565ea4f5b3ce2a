// abi_check.cpp -- test program (not product code): walks the C ABI of include/rt_mi355.h on a GPU with the
// library's HOST code built under AddressSanitizer (scripts/asan_driver.sh): uploads by every route, every
// kind of frame, lists that change between calls, frames in flight, the upscale, options and errors.  It
// checks statuses and that repeated calls give the same bytes; what the bytes must be is the parity tests'
// business.  Needs an MI355X.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <vector>

#include "../../include/rt_mi355.h"

namespace {
int g_errors = 0;
rt_ctx* g_ctx = nullptr;
void check(int st, const char* what) {
    if (st == RT_OK) return;
    if (g_errors++ < 20) fprintf(stderr, "FAILED %s: %s (%s)\n", what, rt_status_string(st), g_ctx ? rt_last_error(g_ctx) : "");
}
void expect(bool ok, const char* what) { if (!ok && g_errors++ < 20) fprintf(stderr, "FAILED: %s\n", what); }

std::vector<rt_triangle> soup(int n, unsigned seed, float spread) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(-spread, spread), e(-0.15f, 0.15f), t(0.0f, 1.0f);
    std::vector<rt_triangle> v((size_t)n);
    for (auto& x : v) {
        memset(&x, 0, sizeof x);
        x.v0 = rt_vec3{u(rng), 0.6f + 0.3f * u(rng), u(rng)};
        x.v1 = rt_vec3{x.v0.x + e(rng), x.v0.y + e(rng), x.v0.z + e(rng)};
        x.v2 = rt_vec3{x.v0.x + e(rng), x.v0.y + e(rng), x.v0.z + e(rng)};
        x.t0 = rt_vec2{t(rng), t(rng)}; x.t1 = rt_vec2{t(rng), t(rng)}; x.t2 = rt_vec2{t(rng), t(rng)};
        x.n0 = x.n1 = x.n2 = rt_vec3{0.0f, 1.0f, 0.0f};
    }
    return v;
}

template <class T> T* dev(size_t n) { void* p = nullptr; if (hipMalloc(&p, n * sizeof(T) + 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(3); } return (T*)p; }
template <class T> std::vector<T> back(const T* d, size_t n) { std::vector<T> h(n); if (hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) exit(3); return h; }
}  // namespace

int main() {
    check(rt_ctx_create(0, &g_ctx), "rt_ctx_create");
    if (!g_ctx) return 2;
    rt_ctx* c = g_ctx;
    std::vector<uint8_t> tex(16 * 16 * 4);
    for (size_t i = 0; i < tex.size(); i++) tex[i] = (uint8_t)(40 + (i * 37) % 200);
    const rt_camera cam = {{-3, 3, -3}, {0, 0, 0}, {0, 1, 0}, 60.0f};
    const rt_light lights[4] = {{{0.57735f, 0.57735f, -0.57735f}, {1.4f, 1.4f, 1.4f}}, {{-0.4f, 0.8f, 0.4f}, {0.3f, 0.5f, 0.9f}},
                                {{0.2f, 0.9f, 0.3f}, {0.5f, 0.2f, 0.2f}}, {{-0.5f, 0.7f, -0.5f}, {0.2f, 0.4f, 0.2f}}};

    // ---- uploads by every route
    int32_t id = -1, n_nodes = 0;
    auto a = soup(700, 1, 1.5f);
    std::vector<rt_flat_node> nodes(2 * a.size());
    check(rt_bvh_build(a.data(), (int32_t)a.size(), nodes.data(), (int32_t)nodes.size(), &n_nodes), "rt_bvh_build");
    check(rt_mesh_upload(c, a.data(), (int32_t)a.size(), nodes.data(), n_nodes, tex.data(), 16, 16, &id), "rt_mesh_upload");
    expect(id == 0, "first mesh id");
    auto b = soup(300, 2, 0.8f);
    std::vector<rt_triangle> b_out(b.size());
    check(rt_mesh_upload_soup(c, b.data(), (int32_t)b.size(), tex.data(), 16, 16, &id, b_out.data()), "rt_mesh_upload_soup (host build)");
    check(rt_set_option(c, "device_build_min", 100), "device_build_min");
    auto g = soup(5000, 3, 2.5f);
    for (auto& t : g) { t.v0.y *= 0.05f; t.v1.y = t.v0.y; t.v2.y = t.v0.y; }             // a flat sheet: the ground
    check(rt_mesh_upload_soup(c, g.data(), (int32_t)g.size(), tex.data(), 16, 16, &id, nullptr), "rt_mesh_upload_soup (device build)");
    expect(id == 2 && rt_mesh_count(c) == 3, "three meshes");
    std::vector<rt_flat_node> gn(2 * g.size());
    auto g2 = g;
    check(rt_bvh_build_device(c, g2.data(), (int32_t)g2.size(), gn.data(), (int32_t)gn.size(), &n_nodes), "rt_bvh_build_device");
    float ms = 0; int32_t lv = 0, ln = 0;
    check(rt_bvh_build_stats(c, &ms, &lv, &ln), "rt_bvh_build_stats");
    check(rt_mesh_replace(c, 2, g2.data(), (int32_t)g2.size(), gn.data(), n_nodes), "rt_mesh_replace");
    expect(rt_bvh_build_device(c, g2.data(), (int32_t)g2.size(), gn.data(), 3, &n_nodes) != RT_OK, "node array too small refused");

    // ---- frames
    const int W = 203, H = 117;
    const size_t px = (size_t)W * H;
    rt_transform tf[3] = {{{0.5f, 0.4f, 0.2f}, {0.1f, 0.7f, 0}}, {{-0.8f, 0.2f, 0.6f}, {0, 2.0f, 0.1f}}, {{0, 0, 0}, {0, 0, 0}}};
    std::vector<uint8_t> ref(px * 3), h_rgb(px * 3), h_sh(px);
    std::vector<int32_t> h_id(px);
    std::vector<float> h_t(px);
    check(rt_render_frame_host(c, &cam, &lights[0], tf, 3, W, H, 0, ref.data(), h_id.data(), h_t.data(), h_sh.data()), "rt_render_frame_host");
    size_t hits = 0; for (int32_t v : h_id) hits += v >= 0;
    expect(hits > px / 10, "the test scene is visible");
    uint8_t* d_rgb = dev<uint8_t>(px * 3 * 16 + 16 * 64);
    int32_t* d_id = dev<int32_t>(px);
    int32_t* d_tri = dev<int32_t>(px);
    float* d_t = dev<float>(px);
    uint8_t* d_sh = dev<uint8_t>(px);
    int nv = 0;
    check(rt_get_option(c, "n_variants", &nv), "n_variants");
    for (int v = 0; v < nv; v++)
        for (int blk : {64, 256}) {
            check(rt_set_option(c, "variant", v), "variant");
            check(rt_set_option(c, "block", blk), "block");
            check(rt_render_frame(c, &cam, &lights[0], tf, 3, W, H, 0, d_rgb, d_id, d_t, d_sh, nullptr), "rt_render_frame");
            check(rt_sync(c, nullptr), "rt_sync");
            expect(back(d_rgb, px * 3) == ref, "every variant gives the same frame");
        }
    check(rt_set_option(c, "variant", 3), "variant"); check(rt_set_option(c, "block", 64), "block");
    for (uint32_t flags : {(uint32_t)RT_FLAG_COUNT_WORK, (uint32_t)(RT_FLAG_COUNT_WORK | RT_FLAG_REFERENCE_ORDER)}) {
        check(rt_render_frame(c, &cam, &lights[0], tf, 3, W, H, flags, d_rgb, nullptr, nullptr, nullptr, nullptr), "rt_render_frame counting");
        rt_counters k;
        check(rt_last_counters(c, &k), "rt_last_counters");
        expect(k.primary_rays == px && k.hit_pixels == hits, "counters");
        expect(back(d_rgb, px * 3) == ref, "counting frame is the same frame");
    }
    {   // the kernels check their own shortcuts: same frame, nothing contradicted
        check(rt_render_frame(c, &cam, &lights[0], tf, 3, W, H, RT_FLAG_VERIFY_SKIPS, d_rgb, nullptr, nullptr, nullptr, nullptr), "rt_render_frame self-check");
        rt_counters k;
        check(rt_last_counters(c, &k), "rt_last_counters");
        expect(k.primary_rays == px && k.hit_pixels == hits && k.skip_checks >= px, "self-check examined every query");
        expect(k.vis_disagreements + k.shadow_vis_disagreements + k.entry_disagreements + k.cull_disagreements + k.instance_disagreements == 0,
               "no shortcut contradicted by the reference's way");
        expect(back(d_rgb, px * 3) == ref, "self-checking frame is the same frame");
    }
    for (int n : {1, 5, 16}) {
        std::vector<rt_transform> many((size_t)n * 3);
        for (int f = 0; f < n; f++) for (int m = 0; m < 3; m++) many[(size_t)f * 3 + m] = tf[m];
        const size_t stride = px * 3 + 64;
        check(rt_render_frames(c, &cam, &lights[0], many.data(), n, 3, W, H, d_rgb, stride, nullptr), "rt_render_frames");
        check(rt_sync(c, nullptr), "rt_sync");
        expect(back(d_rgb + (size_t)(n - 1) * stride, px * 3) == ref, "the last frame of a batch is the same frame");
    }
    expect(rt_render_frames(c, &cam, &lights[0], tf, 17, 3, W, H, d_rgb, px * 3, nullptr) != RT_OK, "17 frames refused");
    // strips put together = the frame
    for (int stride : {2, 3}) {
        std::vector<uint8_t> whole(px * 3, 0);
        for (int off = 0; off < stride; off++) {
            const int rows = rt_strip_rows(H, stride, off);
            check(rt_render_frame_strips(c, &cam, &lights[0], tf, 3, W, H, stride, off, 0, d_rgb, nullptr, nullptr, nullptr, nullptr), "rt_render_frame_strips");
            check(rt_sync(c, nullptr), "rt_sync");
            auto part = back(d_rgb, (size_t)rows * W * 3);
            int r = 0;
            for (int s = off; s * 8 < H; s += stride)
                for (int y = s * 8; y < s * 8 + 8 && y < H; y++, r++) memcpy(&whole[(size_t)y * W * 3], &part[(size_t)r * W * 3], (size_t)W * 3);
            expect(r == rows, "strip rows");
        }
        expect(whole == ref, "strips put together are the frame");
    }
    // sampled frames
    check(rt_render_frame_sampled(c, &cam, lights, 1, tf, 3, W, H, 1, 0, 0, d_rgb, d_id, nullptr), "rt_render_frame_sampled 1x1");
    check(rt_sync(c, nullptr), "rt_sync");
    expect(back(d_rgb, px * 3) == ref, "one sample, one light is the plain frame");
    std::vector<uint8_t> s16;
    for (int split : {0, 2, 1})
        for (int spp : {4, 16, 64}) {
            check(rt_set_option(c, "sampled_split", split), "sampled_split");
            check(rt_render_frame_sampled(c, &cam, lights, 4, tf, 3, W, H, spp, 7, 0, d_rgb, nullptr, nullptr), "rt_render_frame_sampled");
            check(rt_sync(c, nullptr), "rt_sync");
            if (spp == 16) { auto now = back(d_rgb, px * 3); if (s16.empty()) s16 = now; expect(now == s16, "looped and split sampled frames agree"); }
        }
    check(rt_render_frame_sampled_strips(c, &cam, lights, 2, tf, 3, W, H, 8, 7, 4, 1, RT_FLAG_COUNT_WORK, d_rgb, d_id, nullptr), "rt_render_frame_sampled_strips");
    check(rt_sync(c, nullptr), "rt_sync");
    expect(rt_render_frame_sampled(c, &cam, lights, 5, tf, 3, W, H, 4, 0, 0, d_rgb, nullptr, nullptr) != RT_OK, "five lights refused");

    // ---- a mesh rebuilt between frames, by both routes
    for (int round = 0; round < 3; round++) {
        auto moved = soup(300 + 40 * round, 10 + round, 0.8f);
        check(rt_mesh_rebuild(c, 1, moved.data(), (int32_t)moved.size(), nullptr), "rt_mesh_rebuild (device)");
        check(rt_render_frame(c, &cam, &lights[0], tf, 3, W, H, 0, d_rgb, nullptr, nullptr, nullptr, nullptr), "rt_render_frame after rebuild");
    }
    check(rt_set_option(c, "device_build_min", 1000000), "device_build_min");
    check(rt_mesh_rebuild(c, 1, b.data(), (int32_t)b.size(), b_out.data()), "rt_mesh_rebuild (host)");
    check(rt_render_frame_host(c, &cam, &lights[0], tf, 3, W, H, 0, h_rgb.data(), nullptr, nullptr, nullptr), "rt_render_frame_host");
    expect(h_rgb == ref, "the scene is back where it was");

    // ---- instanced lists that change from call to call
    {
        std::mt19937 rng(5);
        std::uniform_real_distribution<float> u(-6.0f, 6.0f), ang(0.0f, 6.28f);
        std::vector<rt_instance> inst(400);
        for (size_t i = 0; i < inst.size(); i++) inst[i] = rt_instance{(int32_t)(i % 2), {{u(rng), 0.2f, u(rng)}, {0, ang(rng), 0}}};
        inst.push_back(rt_instance{2, {{0, 0, 0}, {0, 0, 0}}});
        std::vector<uint8_t> first(px * 3), now(px * 3);
        check(rt_render_frame_instanced_host(c, &cam, &lights[0], inst.data(), (int32_t)inst.size(), W, H, 0, first.data()), "rt_render_frame_instanced_host");
        const auto keep = inst;
        for (int step = 0; step < 12; step++) {
            const int movers = step % 4 == 3 ? 300 : 1 + step;
            for (int k = 0; k < movers; k++) { auto& x = inst[rng() % 400]; x.transform.position = rt_vec3{u(rng), 0.2f, u(rng)}; if (k % 5 == 0) x.mesh ^= 1; }
            check(rt_render_frame_instanced(c, &cam, &lights[0], inst.data(), (int32_t)inst.size(), W, H, step & 1 ? RT_FLAG_COUNT_WORK : 0, d_rgb, d_id, d_tri,
                                            d_t, d_sh, nullptr), "rt_render_frame_instanced");
        }
        check(rt_render_frame_instanced(c, &cam, &lights[0], inst.data(), 10, W, H, 0, d_rgb, nullptr, nullptr, nullptr, nullptr, nullptr), "shorter list");
        check(rt_render_frame_instanced(c, &cam, &lights[0], nullptr, 0, W, H, 0, d_rgb, nullptr, nullptr, nullptr, nullptr, nullptr), "empty list");
        check(rt_render_frame_instanced_host(c, &cam, &lights[0], keep.data(), (int32_t)keep.size(), W, H, 0, now.data()), "rt_render_frame_instanced_host");
        expect(now == first, "the first list again gives the first frame");
        check(rt_set_option(c, "keep_hierarchy", 0), "keep_hierarchy");
        check(rt_render_frame_instanced_host(c, &cam, &lights[0], keep.data(), (int32_t)keep.size(), W, H, 0, now.data()), "rt_render_frame_instanced_host");
        expect(now == first, "and so does a hierarchy built anew");
        check(rt_set_option(c, "keep_hierarchy", 1), "keep_hierarchy");
        rt_instance bad = {7, {{0, 0, 0}, {0, 0, 0}}};
        expect(rt_render_frame_instanced(c, &cam, &lights[0], &bad, 1, W, H, 0, d_rgb, nullptr, nullptr, nullptr, nullptr, nullptr) != RT_OK, "unknown mesh refused");
    }

    // ---- frames in flight, pinned and pageable destinations
    {
        void* pin = nullptr;
        check(rt_pinned_alloc(px * 3 * 6, &pin), "rt_pinned_alloc");
        std::vector<uint8_t> pageable(px * 3 * 6);
        check(rt_set_option(c, "frames_in_flight", 3), "frames_in_flight");
        for (int f = 0; f < 6; f++) {
            check(rt_render_frame_host(c, &cam, &lights[0], tf, 3, W, H, 0, (uint8_t*)pin + (size_t)f * px * 3, nullptr, nullptr, nullptr), "frame in flight (pinned)");
            check(rt_render_frame_host(c, &cam, &lights[0], tf, 3, W, H, 0, pageable.data() + (size_t)f * px * 3, nullptr, nullptr, nullptr), "frame in flight (pageable)");
        }
        check(rt_sync(c, nullptr), "rt_sync");
        for (int f = 0; f < 6; f++)
            expect(memcmp((uint8_t*)pin + (size_t)f * px * 3, ref.data(), px * 3) == 0 && memcmp(pageable.data() + (size_t)f * px * 3, ref.data(), px * 3) == 0, "frames in flight arrive");
        check(rt_set_option(c, "frames_in_flight", 1), "frames_in_flight");
        check(rt_pinned_free(pin), "rt_pinned_free");
    }

    // ---- upscale
    {
        int32_t sw = 0, sh = 0;
        check(rt_scaled_size(W, H, 0.9f, &sw, &sh), "rt_scaled_size");
        std::vector<uint32_t> up((size_t)sw * sh), up2;
        check(rt_upscale_frame_host(c, ref.data(), W, H, sw, sh, up.data()), "rt_upscale_frame_host");
        uint32_t* d_up = dev<uint32_t>((size_t)sw * sh * 3);
        if (hipMemcpy(d_rgb, ref.data(), px * 3, hipMemcpyHostToDevice) != hipSuccess) return 3;
        if (hipMemcpy(d_rgb + px * 3, ref.data(), px * 3, hipMemcpyHostToDevice) != hipSuccess) return 3;
        check(rt_upscale_frames(c, d_rgb, px * 3, 2, W, H, sw, sh, d_up, (size_t)sw * sh, nullptr), "rt_upscale_frames");
        check(rt_sync(c, nullptr), "rt_sync");
        expect(back(d_up + (size_t)sw * sh, (size_t)sw * sh) == up, "device and host upscale agree");
        (void)hipFree(d_up);
    }

    // ---- the end: an empty scene renders the background
    check(rt_scene_clear(c), "rt_scene_clear");
    check(rt_render_frame_host(c, &cam, &lights[0], nullptr, 0, W, H, 0, h_rgb.data(), nullptr, nullptr, nullptr), "empty scene");
    expect(h_rgb[0] == 50 && h_rgb[px * 3 - 1] == 50, "background");
    expect(rt_set_option(c, "no_such_option", 1) != RT_OK, "unknown option refused");
    (void)hipFree(d_rgb); (void)hipFree(d_id); (void)hipFree(d_tri); (void)hipFree(d_t); (void)hipFree(d_sh);
    check(rt_ctx_destroy(c), "rt_ctx_destroy");
    printf("{\"errors\": %d}\n", g_errors);
    return g_errors ? 1 : 0;
}
