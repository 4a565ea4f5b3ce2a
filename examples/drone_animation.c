/* drone_animation.c -- a C program on the reference's host API (scene.h), linked against
 * libraytracer_host.so instead of the reference's own objects.
 *
 * It makes the calls the reference's driver makes, in the same order (raytracer.c:8-75): a scene of
 * 800x600 at scale 0.9 (720x540 rendered), 4 s at 24 frames per second, camera, one light, the three
 * meshes, per frame the placements of raytracer.c:41-53, render_scene + next_frame, then save_scene.
 * Nothing in it names a GPU: which device renders is the library's business.
 *
 *   drone_animation <assets dir> <out.webp> [frames in flight] [ppm prefix] [gpus]
 *
 * With a ppm prefix ("-" for none), frames 0, 45 and 95 are also written as <prefix>_<frame>.ppm.  With a GPU count
 * above one, frame f is rendered on GPU f mod gpus (use it with two or more frames in flight).
 * Build: gcc -std=c11 -O2 -I raytracer_amd/host examples/drone_animation.c \
 *            -L raytracer_amd/host -lraytracer_host -Wl,-rpath,$PWD/raytracer_amd/host -lm
 */
#include "raytracer_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static double seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <assets dir> <out.webp> [frames in flight] [ppm prefix] [gpus]\n", argv[0]);
        return 2;
    }
    const char* assets = argv[1];
    const char* out = argv[2];
    const int in_flight = argc > 3 ? atoi(argv[3]) : 1;
    const char* ppm = argc > 4 && argv[4][0] != '-' ? argv[4] : NULL;
    const int gpus = argc > 5 ? atoi(argv[5]) : 1;
    if (gpus > 1) {
        int list[64];
        if (gpus > 64) { fprintf(stderr, "at most 64 GPUs\n"); return 2; }
        for (int i = 0; i < gpus; i++) list[i] = i;
        if (rt_host_set_gpus(list, gpus) != 0) { fprintf(stderr, "this machine does not have %d GPUs\n", gpus); return 1; }
    }

    Scene scene = create_scene(800, 600, 4000, 24, 0.9f);
    set_scene_camera(&scene, (Vec3){-3.0f, 3.0f, -3.0f}, (Vec3){0.0f, 0.0f, 0.0f}, (Vec3){0.0f, 1.0f, 0.0f}, 60.0f);
    set_scene_light(&scene, (Vec3){1.0f, 1.0f, -1.0f}, (Vec3){1.4f, 1.4f, 1.4f});

    static const char* names[3] = {"drone", "treasure", "ground"};
    Mesh meshes[3];
    int loaded = 0, status = 0;
    for (int i = 0; i < 3; i++) {
        char obj[1024], tex[1024];
        snprintf(obj, sizeof obj, "%s/%s.obj", assets, names[i]);
        snprintf(tex, sizeof tex, "%s/%s.webp", assets, names[i]);
        meshes[i] = create_mesh(obj, tex);
        loaded = i + 1;
        if (meshes[i].triangle_count == 0) { fprintf(stderr, "cannot load %s\n", obj); status = 1; goto done; }
        add_mesh_to_scene(&scene, meshes[i]);
    }
    if (in_flight > 1) rt_host_set_frames_in_flight(in_flight);

    const double t0 = seconds();
    for (int frame = 0; frame < scene.frame_count; frame++) {
        const float t = frame * (2.0f * M_PI / 120.0f);
        set_mesh_position(&scene.meshes[0], (Vec3){2.0f * cosf(t), 1.0f + 0.2f * sinf(2 * t), 2.0f * sinf(t)});
        set_mesh_rotation(&scene.meshes[0], (Vec3){0.1f * sinf(t), t, 0.1f * cosf(t)});
        set_mesh_position(&scene.meshes[1], (Vec3){1.0f, 0.5f + 0.1f * sinf(t), 1.0f});
        set_mesh_rotation(&scene.meshes[1], (Vec3){0, t * 0.5f, 0});
        render_scene(&scene);
        if (rt_host_last_status(&scene) != 0) { fprintf(stderr, "frame %d failed\n", frame); status = 1; goto done; }
        next_frame(&scene);
    }
    if (rt_host_wait(&scene) != 0) { status = 1; goto done; }
    const double t1 = seconds();
    if (ppm) {
        static const int which[3] = {0, 45, 95};
        for (int i = 0; i < 3; i++) {
            char name[1024];
            snprintf(name, sizeof name, "%s_%d.ppm", ppm, which[i]);
            if (which[i] < scene.frame_count && rt_host_write_ppm(&scene, which[i], name) != 0) { status = 1; goto done; }
        }
    }
    save_scene(&scene, out);
    const double t2 = seconds();
    printf("{\"frames\": %d, \"width\": %d, \"height\": %d, \"frames_in_flight\": %d, \"gpus\": %d, \"render_s\": %.4f, \"save_s\": %.4f}\n",
           scene.frame_count, scene.width, scene.height, in_flight, gpus, t1 - t0, t2 - t1);

done:
    for (int i = 0; i < loaded; i++) destroy_mesh(&meshes[i]);
    destroy_scene(&scene);
    return status;
}
