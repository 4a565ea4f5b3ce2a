// host_check.cpp -- test program (not product code): the CPU side of the host library (OBJ reader in both
// modes, texture loading, BVH build, the animation writer) compiled together with it under the address and
// undefined-behaviour sanitizers and driven with well-formed, malformed and randomly damaged inputs.
// Nothing here renders; no GPU is needed.  Built and run by tests/test_abi_and_host.py.
//
//   host_check <assets dir> <scratch dir>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

#include "../../raytracer_amd/host/raytracer_host.h"

namespace {

int g_errors = 0;
void expect(bool ok, const char* what) { if (!ok && g_errors++ < 20) fprintf(stderr, "FAILED: %s\n", what); }

std::string slurp(const std::string& path) {
    std::string s;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return s;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

void spill(const std::string& path, const std::string& data) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { expect(false, "cannot write scratch file"); return; }
    fwrite(data.data(), 1, data.size(), f);
    fclose(f);
}

// Loads, looks at every triangle and BVH node, frees.  Returns the triangle count.
size_t count_nodes(const BVHNode* n, size_t total, size_t& leaf_tris) {
    if (!n) return 0;
    if (!n->left && !n->right) {
        expect(n->start_idx >= 0 && n->triangle_count >= 0 && (size_t)n->start_idx + (size_t)n->triangle_count <= total, "leaf range inside the triangle array");
        leaf_tris += (size_t)n->triangle_count;
        return 1;
    }
    return 1 + count_nodes(n->left, total, leaf_tris) + count_nodes(n->right, total, leaf_tris);
}

size_t load(const std::string& obj, const std::string& tex) {
    Mesh m = create_mesh(obj.c_str(), tex.c_str());
    double sum = 0.0;
    for (size_t i = 0; i < m.triangle_count; i++) sum += m.triangles[i].v0.x + m.triangles[i].t2.v + m.triangles[i].n1.z;
    size_t leaf_tris = 0;
    count_nodes(m.bvh.root, m.triangle_count, leaf_tris);
    // (a mesh whose texture could not be loaded comes back without a BVH, mesh.c:82-88)
    if (m.bvh.root && leaf_tris != m.triangle_count) {
        fprintf(stderr, "%s: %zu triangles, %zu in leaves\n", obj.c_str(), m.triangle_count, leaf_tris);
        expect(false, "every triangle in exactly one leaf");
    }
    if (m.texture_data && m.texture_width > 0 && m.texture_height > 0) {
        const Vec3 c = sample_mesh_texture(&m, 0.3f, -2.6f);
        expect(c.x >= 0.0f && c.x <= 1.0f, "texel in range");
    }
    const size_t n = m.triangle_count;
    (void)sum;
    destroy_mesh(&m);
    return n;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: host_check <assets dir> <scratch dir>\n"); return 2; }
    const std::string assets = argv[1], tmp = argv[2];
    const std::string tex = assets + "/drone.webp";

    // ---- the reference's assets, both reader modes
    static const char* names[3] = {"drone", "treasure", "ground"};
    static const size_t tris[3] = {308, 3155, 162};
    for (int mode = 0; mode < 2; mode++) {
        rt_host_set_obj_mode(mode);
        for (int i = 0; i < 3; i++)
            expect(load(assets + "/" + names[i] + ".obj", assets + "/" + names[i] + ".webp") == tris[i], "asset triangle count");
    }

    // ---- malformed files
    const std::string cube_v = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 1\n";
    std::vector<std::string> cases = {
        "",                                                            // empty
        "v 1 2 3",                                                     // no newline, no faces
        cube_v + "f 1/1/1 2/2/1 3/3/1",                                // last line without newline
        cube_v + "f 1/1/1 2/2/1 9/3/1\nf 1/1/1 2/9/1 3/3/1\nf 1/1/1 2/2/9 3/3/1\n",     // indices past the lists
        cube_v + "f 0/0/0 1/1/1 2/2/1\n",                              // index 0
        cube_v + "f -1/-1/-1 -2/-2/-1 -3/-3/-1\nf -9/-1/-1 -2/-2/-1 -3/-3/-1\n",       // negative, and too negative
        cube_v + "f 1/1/1 2/2/1\nf 1/1/1\nf\n",                        // too few corners
        cube_v + "f 1/1/1 2/2/1 3/3/1 4/1/1\nf 1/1/1 2/2/1 3/3/1 4/1/1 1/1/1 2/2/1\n", // quads and more
        cube_v + "f 1 2 3\nf 1/1 2/2 3/3\nf 1//1 2//1 3//1\nf 1/ 2/ 3/\nf 1// 2// 3//\n",
        cube_v + "f 99999999999999999999/1/1 2/2/1 3/3/1\nf 1/1/1 2/2/1 3/3/-99999999999999999999\n",
        cube_v + "f a/b/c d/e/f g/h/i\nf 1/1/1 x 3/3/1\nf 1.5/1/1 2/2/1 3/3/1\n",
        "v 1\nv\nvt\nvn 1 2\nv nan inf -inf\nv 1e999 1e-999 0\nf 1/1/1 2/2/1 3/3/1\n",
        cube_v + "f 1/1/1 2/2/1 3/3/1 " + std::string(300000, '7') + "\n",              // a very long line
        std::string(100000, 'v') + "\n" + cube_v + "f 1/1/1 2/2/1 3/3/1\n",
        std::string("v 0 0 0") + '\0' + "v 1 1 1\nf 1/1/1 1/1/1 1/1/1\n",   // embedded NUL
        cube_v + "\r\nf 1/1/1 2/2/1 3/3/1\r\n# comment\r\ns off\r\nusemtl x\r\n",
    };
    for (int mode = 0; mode < 2; mode++) {
        rt_host_set_obj_mode(mode);
        for (size_t k = 0; k < cases.size(); k++) {
            const std::string path = tmp + "/case.obj";
            spill(path, cases[k]);
            const size_t n = load(path, tex);
            expect(n <= 16, "malformed file gives few triangles");
        }
    }
    // degenerate geometry: many triangles with one centroid (the partition fails: one large leaf)
    {
        std::string s = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\n";
        for (int i = 0; i < 500; i++) s += "f 1/1/1 2/1/1 3/1/1\n";
        spill(tmp + "/same.obj", s);
        rt_host_set_obj_mode(0);
        expect(load(tmp + "/same.obj", tex) == 500, "500 coincident triangles");
    }
    // textures: missing, empty, garbage, truncated
    {
        const std::string obj = assets + "/ground.obj";
        load(obj, tmp + "/no_such.webp");
        spill(tmp + "/empty.webp", "");
        load(obj, tmp + "/empty.webp");
        spill(tmp + "/junk.webp", std::string(5000, 'x'));
        load(obj, tmp + "/junk.webp");
        const std::string good = slurp(tex);
        expect(good.size() > 1000, "texture asset readable");
        spill(tmp + "/cut.webp", good.substr(0, good.size() / 3));
        load(obj, tmp + "/cut.webp");
        load(tmp + "/no_such.obj", tex);
    }
    // ---- randomly damaged copies of a real file
    {
        const std::string good = slurp(assets + "/drone.obj");
        expect(good.size() > 1000, "obj asset readable");
        std::mt19937 rng(7);
        static const char repl[] = "0123456789/-. \n\0fvtn#e+";
        for (int round = 0; round < 120; round++) {
            std::string s = good;
            const int edits = 1 + (int)(rng() % 40);
            for (int e = 0; e < edits; e++) {
                const size_t at = rng() % s.size();
                switch (rng() % 4) {
                    case 0: s[at] = repl[rng() % (sizeof repl - 1)]; break;
                    case 1: s.erase(at, 1 + rng() % 30); break;
                    case 2: s.insert(at, std::string(1 + rng() % 5, repl[rng() % (sizeof repl - 1)])); break;
                    default: s.resize(at + 1); break;
                }
                if (s.empty()) s = "f";
            }
            spill(tmp + "/fuzz.obj", s);
            rt_host_set_obj_mode(round & 1);
            load(tmp + "/fuzz.obj", round % 10 == 0 ? tex : tmp + "/no_such.webp");
        }
        rt_host_set_obj_mode(0);
    }
    // ---- BVH build on its own: random, tiny and degenerate inputs
    {
        std::mt19937 rng(3);
        std::uniform_real_distribution<float> u(-5.0f, 5.0f);
        for (size_t n : {(size_t)0, (size_t)1, (size_t)4, (size_t)5, (size_t)1000}) {
            std::vector<Triangle> t(n ? n : 1);
            memset(t.data(), 0, t.size() * sizeof(Triangle));
            for (size_t i = 0; i < n; i++) {
                t[i].v0 = Vec3{u(rng), u(rng), u(rng)};
                t[i].v1 = Vec3{t[i].v0.x + 0.1f, t[i].v0.y, t[i].v0.z};
                t[i].v2 = Vec3{t[i].v0.x, t[i].v0.y + 0.1f, t[i].v0.z};
            }
            if (n >= 5) { t[2].v0.x = NAN; t[3].v1.y = INFINITY; }
            BVH b = create_bvh(t.data(), n);
            size_t leaf_tris = 0;
            count_nodes(b.root, n, leaf_tris);
            expect(leaf_tris == n, "bvh covers its triangles");
            destroy_bvh(&b);
        }
    }
    // ---- the animation writer: frames in both formats, odd sizes, nothing added, bad arguments
    {
        rt_host_animation* a = nullptr;
        expect(rt_host_animation_open((tmp + "/a.webp").c_str(), 37, 21, 60.0f, 3, &a) == 0 && a, "animation opens");
        std::vector<unsigned char> rgb(37 * 21 * 3);
        std::vector<uint32_t> argb(37 * 21);
        for (int f = 0; f < 9; f++) {
            for (size_t i = 0; i < rgb.size(); i++) rgb[i] = (unsigned char)(i * 7 + f * 31);
            for (size_t i = 0; i < argb.size(); i++) argb[i] = 0xFF000000u | (uint32_t)(i * 2654435761u + f);
            expect(rt_host_animation_add(a, f & 1 ? (const void*)argb.data() : (const void*)rgb.data(), f & 1, 40 + f) == 0, "frame added");
        }
        expect(rt_host_animation_add(a, nullptr, 0, 40) != 0, "null frame refused");
        expect(rt_host_animation_add(a, rgb.data(), 7, 40) != 0, "unknown format refused");
        expect(rt_host_animation_close(a) == 0, "animation closes");
        a = nullptr;
        expect(rt_host_animation_open((tmp + "/b.webp").c_str(), 5, 3, -1.0f, 1, &a) == 0 && a, "lossless animation opens");
        expect(rt_host_animation_close(a) == 0, "animation without frames closes");
        a = nullptr;
        expect(rt_host_animation_open((tmp + "/no_such_dir/c.webp").c_str(), 5, 3, 75.0f, 1, &a) != 0 && !a, "unwritable path refused");
        expect(rt_host_animation_open((tmp + "/d.webp").c_str(), 0, 3, 75.0f, 1, &a) != 0, "zero width refused");
        expect(rt_host_animation_close(nullptr) != 0, "null handle refused");
    }
    // ---- a scene without a GPU behind it: every call returns, nothing is rendered
    {
        Scene s = create_scene(33, 17, 500, 10, 0.7f);
        set_scene_camera(&s, Vec3{-3, 3, -3}, Vec3{0, 0, 0}, Vec3{0, 1, 0}, 60.0f);
        set_scene_light(&s, Vec3{1, 1, -1}, Vec3{1.4f, 1.4f, 1.4f});
        Mesh m = create_mesh((assets + "/ground.obj").c_str(), (assets + "/ground.webp").c_str());
        add_mesh_to_scene(&s, m);
        for (int f = 0; f < s.frame_count + 2; f++) next_frame(&s);
        expect(s.current_frame < s.frame_count, "next_frame stays inside the frame list");
        destroy_mesh(&m);
        destroy_scene(&s);
    }
    printf("{\"errors\": %d}\n", g_errors);
    return g_errors ? 1 : 0;
}
