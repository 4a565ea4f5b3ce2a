/*
 * rt_mi355.h -- C ABI of the MI355X rendering core (librt_mi355.so).
 *
 * This is the drop-in boundary for the reference ray tracer's per-pixel ray/scene loop.
 * The reference has no FFI of its own: its public C headers are the interface, and its
 * driver calls render_scene(Scene*) once per frame (reference raytracer.c:56, scene.h:34).
 * A host that keeps the reference's structs binds the entry points below instead of
 * running scene.c:52-160 on the CPU.  All structs are plain C, layout-identical to the
 * reference structs they are named after, so reference-side pointers can be passed as is.
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_status; nothing aborts.
 * Pointers named d_* are DEVICE pointers (HBM of the context's GPU), h_* are host pointers.
 * A context is bound to one GPU and is not thread-safe; use one context per host thread.
 */
#ifndef RT_MI355_H
#define RT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,       /* bad argument (NULL, negative size, unknown id ...) */
    RT_ERR_DEVICE = -2,        /* a HIP call failed; rt_last_error() has the text */
    RT_ERR_NO_DEVICE = -3,     /* no usable GPU / device index out of range */
    RT_ERR_UNSUPPORTED = -4,   /* input outside the supported envelope (limits below) */
    RT_ERR_CAPACITY = -5       /* caller-provided output array too small */
} rt_status;

/* Limits of this build. */
#define RT_MAX_MESHES        8          /* meshes per scene (one BVH each, reference scene.c:72) in rt_render_frame,
                                           rt_render_frames and rt_render_frame_sampled */
#define RT_MAX_GEOMETRIES    256        /* meshes that can be uploaded (rt_render_frame_instanced places them) */
#define RT_MAX_TREE_DEPTH    30         /* deepest BVH the traversal stack covers */
#define RT_MAX_MESH_TRIS     (1 << 24)  /* triangles per mesh */
#define RT_MAX_BATCH         16         /* frames per rt_render_frames call */
#define RT_MAX_INSTANCES     (1 << 20)  /* placed meshes per rt_render_frame_instanced call */
#define RT_MAX_LIGHTS        4          /* lights per rt_render_frame_sampled call */
#define RT_MAX_SAMPLES       64         /* camera rays per pixel in rt_render_frame_sampled */

/* ---- reference-layout structs ---------------------------------------------------------- */
typedef struct { float x, y, z; } rt_vec3;                       /* math/vec3.h:6 */
typedef struct { float u, v; } rt_vec2;                          /* math/vec3.h:7 */
typedef struct { rt_vec3 position, rotation; } rt_transform;     /* math/ray.h:12-15 (Transform) */
typedef struct { rt_vec3 position, look_at, up; float fov; } rt_camera;   /* render/camera.h:7-12 (Camera) */
typedef struct { rt_vec3 direction, color; } rt_light;           /* render/light.h:6-9 (DirectionalLight);
                                                                    direction already normalised (light.c:4) */
typedef struct {                                                  /* geometry/triangle.h:6-10 (Triangle, 96 B) */
    rt_vec3 v0, v1, v2;
    rt_vec2 t0, t1, t2;
    rt_vec3 n0, n1, n2;
} rt_triangle;

/* accel/bvh.h:8-14 (BVHNode) as the reference lays it out on LP64: a heap node with child
 * pointers.  rt_bvh_flatten_tree() reads a tree of these in place. */
typedef struct rt_tree_node {
    rt_vec3 bounds_min, bounds_max;
    struct rt_tree_node* left;
    struct rt_tree_node* right;
    int start_idx;
    int triangle_count;
} rt_tree_node;

/* The same node with indices instead of pointers (-1 = no child).  nodes[0] is the root;
 * a node is followed by its whole left subtree, then its whole right subtree. */
typedef struct {
    rt_vec3 bounds_min, bounds_max;
    int32_t left, right;
    int32_t start_idx, triangle_count;
} rt_flat_node;

typedef struct rt_ctx rt_ctx;                                     /* opaque */

/* Work counters of one rendering call made with RT_FLAG_COUNT_WORK or RT_FLAG_VERIFY_SKIPS. */
typedef struct {
    uint64_t primary_rays, shadow_rays;     /* rays traced (a ray = one ray against the whole scene) */
    uint64_t slab_tests, tri_tests;         /* ray/AABB and ray/triangle tests executed (RT_FLAG_COUNT_WORK only) */
    uint64_t hit_pixels, shadowed_pixels;
    /* RT_FLAG_VERIFY_SKIPS only (zero otherwise): shortcut decisions that the kernel examined again the reference's way, and
     * how many of them the reference's way contradicts -- all of the following must be zero:
     *   vis_disagreements         a camera ray passes the root box (scene.c:79, bvh.c:82) of a mesh that the screen rectangle of
     *                             its tile leaves out
     *   shadow_vis_disagreements  the same for a shadow ray (scene.c:131) and the shadow rectangle
     *   entry_disagreements       a walk that entered a flat mesh through its grid ends differently from the walk from the root
     *   cull_disagreements        the ordered, distance-culled walk through the second tree ends differently from the
     *                             reference's own walk (its tree, every box whose ancestors pass, nothing culled: bvh.c:80-129)
     *   instance_disagreements    rt_render_frame_instanced: the walk over the instance hierarchy ends differently from the
     *                             reference's loop over the whole mesh list (scene.c:72-108, 124-139) */
    uint64_t skip_checks;
    uint64_t vis_disagreements, shadow_vis_disagreements, entry_disagreements, cull_disagreements, instance_disagreements;
    /* RT_FLAG_COUNT_WORK: how many of shadow_rays start on a surface whose own term max(0.2, n.L) is 0.2 already, so that
     * scene.c:143-144 gives diffuse = 0.2 whatever they meet.  A counting call sends them like the reference; a call that
     * stores the colour only does not (see rt_render_frame). */
    uint64_t idle_shadow_rays;
} rt_counters;

/* ---- context --------------------------------------------------------------------------- */
/* Replaces nothing in the reference (it has no device); pairs with create_scene /
 * destroy_scene (scene.c:8-29, 227-239) in a host binding. */
int rt_ctx_create(int device, rt_ctx** out);
int rt_ctx_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);                     /* text of the last failure ("" if none) */
const char* rt_status_string(int status);

/* ---- BVH build (host) ------------------------------------------------------------------ */
/* create_bvh / create_bvh_node (accel/bvh.c:4-67): longest-axis, mean-of-centroids split,
 * swap-to-front partition, leaves of <= 4 triangles.  Reorders tris IN PLACE exactly as the
 * reference does (triangle indices reported by the renderer are indices into that order)
 * and writes the flat node array.  cap >= 2*n is always enough. */
int rt_bvh_build(rt_triangle* tris, int32_t n, rt_flat_node* nodes_out, int32_t cap, int32_t* n_nodes);
/* The same build on the GPU of ctx, with the same contract and the same bits in tris and nodes_out
 * (create_bvh_node, accel/bvh.c:4-59): all nodes of one depth are processed together -- bounds by a
 * segmented reduction, the centroid mean by one wavefront per node adding in array order (a float sum
 * cannot be reassociated), the swap-to-front partition by its closed form.  Worth it for meshes of
 * 10^5 triangles and more; synchronous.  Trees deeper than 256 levels return RT_ERR_UNSUPPORTED
 * (use rt_bvh_build).  rt_bvh_build_stats: device time (without the copies), tree levels and kernel
 * launches of the last build; any pointer may be NULL. */
int rt_bvh_build_device(rt_ctx* ctx, rt_triangle* tris, int32_t n, rt_flat_node* nodes_out, int32_t cap, int32_t* n_nodes);
int rt_bvh_build_stats(const rt_ctx* ctx, float* device_ms, int32_t* levels, int32_t* launches);
/* Converts a reference pointer tree (Mesh.bvh.root, geometry/mesh.h:17) to the flat form. */
int rt_bvh_flatten_tree(const rt_tree_node* root, rt_flat_node* nodes_out, int32_t cap, int32_t* n_nodes);

/* ---- scene upload ---------------------------------------------------------------------- */
/* Takes what create_mesh (geometry/mesh.c:6-119) leaves in a Mesh: the triangle array in
 * post-build order, its BVH, and the WebPDecodeRGBA texture (row-major RGBA8).  Copies all of
 * it to the GPU in the kernel's own layout; the host arrays may be freed afterwards.
 * Meshes get ids 0,1,2,... in upload order = the order add_mesh_to_scene (scene.c:31) gave them. */
int rt_mesh_upload(rt_ctx* ctx, const rt_triangle* tris, int32_t n_tris,
                   const rt_flat_node* nodes, int32_t n_nodes,
                   const uint8_t* h_rgba, int32_t tex_width, int32_t tex_height, int32_t* mesh_id);
int rt_mesh_upload_tree(rt_ctx* ctx, const rt_triangle* tris, int32_t n_tris, const rt_tree_node* root,
                        const uint8_t* h_rgba, int32_t tex_width, int32_t tex_height, int32_t* mesh_id);
/* Replaces the triangles and BVH of an uploaded mesh (same argument meaning as rt_mesh_upload) and keeps
 * its texture: for a mesh whose vertices change from frame to frame and whose BVH is rebuilt each time
 * (the reference never does this -- its meshes move by Transform only, raytracer.c:44-53 -- but its data
 * structures allow it: rebuild with create_bvh / rt_bvh_build / rt_bvh_build_device, then call this).
 * Waits for frames in flight.  Device arrays are reused when the triangle count is unchanged. */
int rt_mesh_replace(rt_ctx* ctx, int32_t mesh_id, const rt_triangle* tris, int32_t n_tris, const rt_flat_node* nodes, int32_t n_nodes);
/* The same two steps for a triangle soup in file order (what create_mesh has before it calls create_bvh,
 * mesh.c:25-110): builds the reference's BVH -- on the GPU from "device_build_min" triangles up (rt_set_option,
 * default 20 000), records for the kernels made there too, nothing but the nodes returns to the host -- and
 * uploads (rt_mesh_upload_soup) or swaps it into an uploaded mesh (rt_mesh_rebuild, texture kept).  tris_out,
 * if not NULL, receives the n_tris triangles in post-build order, the order reported triangle ids refer to.
 * Results are those of rt_bvh_build + rt_mesh_upload / rt_mesh_replace. */
int rt_mesh_upload_soup(rt_ctx* ctx, const rt_triangle* tris, int32_t n_tris, const uint8_t* h_rgba, int32_t tex_width, int32_t tex_height,
                        int32_t* mesh_id, rt_triangle* tris_out);
int rt_mesh_rebuild(rt_ctx* ctx, int32_t mesh_id, const rt_triangle* tris, int32_t n_tris, rt_triangle* tris_out);
int rt_scene_clear(rt_ctx* ctx);                                  /* drops all meshes */
int rt_mesh_count(const rt_ctx* ctx);

/* ---- rendering ------------------------------------------------------------------------- */
#define RT_FLAG_COUNT_WORK   0x1u   /* also count slab / triangle tests (slower; for measurement) */
/* The default walk visits the nearer child first and skips a box that is entered farther away than the closest hit so far,
 * with a margin of 0.2 % for the rounding of the triangle test's distance.  That margin holds unless a ray meets a triangle's
 * plane at less than about 2e-4 rad, where ray.c:42-67 can put a hit percents in front of the surface; the reference keeps
 * such a hit.  Meshes that hold a triangle with |e1| |e2| >= 0.25 -- large enough for ray.c:50's cut-off to admit such rays
 * over a noticeable area -- are walked without that skip (found and checked with RT_FLAG_VERIFY_SKIPS: 3 rays in 3.6e10
 * before, none in 348e9 checked decisions since).  RT_FLAG_REFERENCE_ORDER is the walk that skips nothing for any mesh. */
#define RT_FLAG_REFERENCE_ORDER 0x2u /* traverse every box the reference would, in its order, no distance culling */
/* Self-check of everything the kernels skip by geometry.  The frame is rendered as without the flag (one lane per ray), but
 * wherever a shortcut is taken the kernel ALSO does what the reference would have done and counts the cases in which that
 * contradicts the shortcut: a mesh left out for a tile by its screen or shadow rectangle gets the reference's root test on
 * the very ray; every query is walked as rendered (second tree, grid entry of flat meshes, nearer child first, distance
 * culling, any-hit shadows), again from the root without the grid, and again as bvh.c:80-129 walks it, and the outcomes are
 * compared; an instanced frame also runs the reference's loop over the whole mesh list per ray.  rt_last_counters reports the
 * result (all *_disagreements must be zero).  Several times slower; overrides RT_FLAG_COUNT_WORK and
 * RT_FLAG_REFERENCE_ORDER; accepted by rt_render_frame[_strips], rt_render_frame_sampled[_strips], rt_render_frame_host and
 * rt_render_frame_instanced[_host] (there the per-ray loop over the list makes it a tool for small frames). */
#define RT_FLAG_VERIFY_SKIPS 0x4u

/* render_scene (scene.c:52-160) for one frame.  transforms[i] is scene->meshes[i].transform
 * (set_mesh_position / set_mesh_rotation, mesh.c:121-127).  Output planes are device pointers:
 *   d_rgb     width*height*3 bytes, row-major RGB8, index (y*width+x)*3      (scene.c:110) -- required
 *   d_hit_id  width*height int32: (mesh_id<<24)|triangle index, -1 = miss   -- optional (NULL)
 *   d_hit_t   width*height float: closest_t, 1e30f = miss                    -- optional
 *   d_shadow  width*height uint8: 1 blocked, 0 lit, 255 = no primary hit     -- optional
 * A call that takes no d_shadow and counts nothing (so also rt_render_frames, rt_render_frame_sampled and the instanced frame
 * without that plane) sends no shadow ray from a surface whose own term max(0.2, n.L) is 0.2 already: scene.c:143-144 then
 * gives diffuse = 0.2 blocked or not, and every byte of the other planes is the same.
 * The launch is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the context's
 * own stream).  Inputs are consumed before the call returns. */
int rt_render_frame(rt_ctx* ctx, const rt_camera* camera, const rt_light* light,
                    const rt_transform* transforms, int32_t n_transforms,
                    int32_t width, int32_t height, uint32_t flags,
                    void* d_rgb, void* d_hit_id, void* d_hit_t, void* d_shadow, void* stream);

/* The reference's frame loop (raytracer.c:47-59: set_mesh_position / set_mesh_rotation, render_scene,
 * next_frame, repeated) for n_frames consecutive frames in ONE launch.  Frames of an animation are
 * independent; rendering them together lets the few long-running tiles of every frame overlap with the
 * bulk of the others, which is what makes the frame rate work-bound.  transforms holds
 * n_frames * n_transforms entries, frame after frame; camera and light are shared.  Frame f is written
 * to (uint8_t*)d_rgb + f * frame_stride (frame_stride >= width*height*3, in bytes), same pixel layout
 * as rt_render_frame, same bytes as n_frames calls of rt_render_frame would give.  Asynchronous on
 * `stream`; the transforms are consumed before the call returns.  1 <= n_frames <= RT_MAX_BATCH. */
int rt_render_frames(rt_ctx* ctx, const rt_camera* camera, const rt_light* light,
                     const rt_transform* transforms, int32_t n_frames, int32_t n_transforms,
                     int32_t width, int32_t height, void* d_rgb, size_t frame_stride, void* stream);

/* rt_render_frames for callers that cannot allocate device memory while they render (tight serving loops, harnesses that
 * forbid hipMalloc in the steady state).  It has no counterpart in the reference.  Same arguments, same launch, same bytes
 * as rt_render_frames with the exact build of the kernels, but the call reaches no allocator: the per-frame constants that
 * rt_render_frames stages in a device buffer of the context go through a file-scope device array of the library
 * (RT_MAX_BATCH records of the per-frame constants, 42 880 bytes, held by every GPU the library runs on until the process ends; its address on the
 * context's GPU is looked up by rt_ctx_create).  Conditions: n_frames <= RT_MAX_BATCH (more: RT_ERR_CAPACITY); the mesh
 * list must not have changed since the last rendering call of this context -- that call put the combined node array on the
 * device; and option "fast" must be off (the entry point promises rt_render_frames' exact bytes) -- otherwise
 * RT_ERR_UNSUPPORTED and nothing is launched.  Eager execution only: the call copies the constants from one of the
 * context's four page-locked staging slots and waits on that slot's event before it reuses it, so it must NOT be made while
 * `stream` is being captured into a HIP graph (a captured copy would read whatever a later call left in the slot, and an
 * event wait is illegal under capture).  All calls on one GPU share the array: calls that overlap on one GPU (different
 * streams, or contexts on the same device) race; calls on one stream are ordered.  Every byte the kernel reads from the array
 * is written by the call itself first. */
int rt_render_frames_noalloc(rt_ctx* ctx, const rt_camera* camera, const rt_light* light,
                             const rt_transform* transforms, int32_t n_frames, int32_t n_transforms,
                             int32_t width, int32_t height, void* d_rgb, size_t frame_stride, void* stream);

/* Beyond the reference (it has one ray through the pixel centre, scene.c:61-62, and one light, scene.h:17):
 * samples_per_pixel jittered camera rays per pixel and n_lights directional lights, every per-ray
 * operation as in render_scene.  Sample s of pixel (x,y) goes through (x+jx, y+jy) with (jx,jy) = (0.5,0.5)
 * when samples_per_pixel is 1 and otherwise two 24-bit fractions from a hash of (x, y, s, seed); every
 * light gets its own shadow ray and contributes (texel*color)*d, d = 0.2 or max(0.2, n.L) when
 * unblocked; a sample's value is min(colour*255, 255) (50 for a miss); the pixel is the mean of its
 * samples' values cast to unsigned char.  With one sample and one light the result equals
 * rt_render_frame's.  lights[i].direction must be normalised (create_directional_light does it).
 * d_hit_id (optional) receives the id hit by sample 0.  With RT_FLAG_COUNT_WORK, rt_last_counters
 * reports primary_rays = pixels*samples, shadow_rays = hit samples*lights, hit_pixels = hit samples,
 * shadowed_pixels = blocked (sample, light) pairs.  RT_FLAG_REFERENCE_ORDER is not supported here. */
int rt_render_frame_sampled(rt_ctx* ctx, const rt_camera* camera, const rt_light* lights, int32_t n_lights,
                            const rt_transform* transforms, int32_t n_transforms,
                            int32_t width, int32_t height, int32_t samples_per_pixel, uint32_t seed,
                            uint32_t flags, void* d_rgb, void* d_hit_id, void* stream);

/* ---- one frame over several GPUs ---------------------------------------------------------- */
/* Pixels of a frame are independent (scene.c:58-59), so a frame can be dealt out in strips of eight rows:
 * the call with (strip_stride, strip_offset) renders strips strip_offset, strip_offset + strip_stride, ...
 * of the frame (0 <= strip_offset < strip_stride) and writes them one below the other into planes that are
 * rt_strip_rows(height, strip_stride, strip_offset) rows tall (the frame's last strip may be shorter than
 * eight rows).  Interleaving balances the load between ranks (sky rows are cheap, rows across a mesh are
 * not); the caller gathers the shares and puts row r of share k at frame row
 * ((r / 8) * strip_stride + k) * 8 + r % 8.  Every pixel is the pixel rt_render_frame /
 * rt_render_frame_sampled would have written.  (1, 0) is the whole frame. */
int rt_strip_rows(int32_t height, int32_t strip_stride, int32_t strip_offset);
int rt_render_frame_strips(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms,
                           int32_t n_transforms, int32_t width, int32_t height, int32_t strip_stride, int32_t strip_offset, uint32_t flags,
                           void* d_rgb, void* d_hit_id, void* d_hit_t, void* d_shadow, void* stream);
int rt_render_frame_sampled_strips(rt_ctx* ctx, const rt_camera* camera, const rt_light* lights, int32_t n_lights, const rt_transform* transforms,
                                   int32_t n_transforms, int32_t width, int32_t height, int32_t samples_per_pixel, uint32_t seed,
                                   int32_t strip_stride, int32_t strip_offset, uint32_t flags, void* d_rgb, void* d_hit_id, void* stream);

/* ---- the animation over several GPUs of one process ---------------------------------------- */
/* The reference's frame loop (raytracer.c:40-57: place the meshes, render_scene, next_frame, for every frame) dealt
 * over n_ctx contexts, normally one per GPU: frame f is rendered by ctxs[f mod n_ctx] (frames are independent: each
 * reads its own placements and writes its own buffer), and all frames are gathered on the GPU of ctxs[0], where
 * save_scene's upscale and encoder consume them (scene.c:186-206): frame f ends up at (uint8_t*)d_rgb + f*frame_stride,
 * d_rgb being memory of ctxs[0]'s GPU that holds n_frames images.  Every context must hold the same meshes (upload the
 * scene to each).  transforms holds n_frames * n_transforms entries, frame after frame.  Batches of up to RT_MAX_BATCH
 * frames go through rt_render_frames on each context's own stream; a batch rendered on another GPU is copied to its
 * place by peer-to-peer copies (xGMI between the GPUs of a node) queued behind its kernel, while that GPU's next batch
 * renders; ctxs[0] renders in place.  No collective is involved: the gather is one point-to-point stream per GPU.
 * The call returns when all frames are in d_rgb, with the calling thread's current device as it found it.  The work runs on
 * the contexts' own streams and is not ordered against anything else: d_rgb must be idle (nothing of the caller's still
 * writing or reading it on any stream) when the call is made.  n_frames may be any number (0: nothing to do); with n_ctx = 1 the
 * result is rt_render_frames' over consecutive batches.  Several contexts on the same GPU are allowed (rehearsal).
 * Errors are reported through ctxs[0] (rt_last_error).  rt_device_count: GPUs visible to the process. */
int rt_render_frames_sharded(rt_ctx* const* ctxs, int32_t n_ctx, const rt_camera* camera, const rt_light* light,
                             const rt_transform* transforms, int32_t n_frames, int32_t n_transforms,
                             int32_t width, int32_t height, void* d_rgb, size_t frame_stride);
int rt_device_count(void);

/* render_scene for a scene whose mesh list is long: the reference's Scene.meshes may hold any number of
 * Mesh values (add_mesh_to_scene, scene.c:31-35), for example the same triangles and BVH under thousands of
 * transforms, and tests every ray against every one of them (scene.c:72-108, 124-139).  Here the uploaded
 * meshes are the distinct geometries and `instances` is the scene's mesh list, in its order: instance i is
 * uploaded mesh instances[i].mesh placed by instances[i].transform.  The result is what render_scene gives
 * for that list (closest hit, the earlier list entry winning equal distances; shadow = any entry blocks).
 * A hierarchy over the instances' world-space boxes, rebuilt on the host for every call, proposes the
 * instances a ray is tested against.  d_hit_instance (optional) receives the list index of the hit
 * (-1 = miss), d_hit_tri (optional) the triangle index inside its mesh, d_hit_t / d_shadow as in
 * rt_render_frame.  Not limited by RT_MAX_MESHES: n_instances <= RT_MAX_INSTANCES over at most
 * RT_MAX_GEOMETRIES uploaded meshes. */
typedef struct { int32_t mesh; rt_transform transform; } rt_instance;
int rt_render_frame_instanced(rt_ctx* ctx, const rt_camera* camera, const rt_light* light,
                              const rt_instance* instances, int32_t n_instances,
                              int32_t width, int32_t height, uint32_t flags,
                              void* d_rgb, void* d_hit_instance, void* d_hit_tri, void* d_hit_t, void* d_shadow,
                              void* stream);

/* The instanced frame into a host buffer (renders into context-owned device memory, waits, copies back). */
int rt_render_frame_instanced_host(rt_ctx* ctx, const rt_camera* camera, const rt_light* light,
                                   const rt_instance* instances, int32_t n_instances,
                                   int32_t width, int32_t height, uint32_t flags, uint8_t* h_rgb);

/* Same, for callers that hold host buffers (what scene->frames[current_frame] is in the
 * reference): renders into context-owned device planes, waits, and copies back.
 * With option "frames_in_flight" = K > 1 (rt_set_option) a call that asks for h_rgb only does not wait: the
 * frame's kernel and its copy to h_rgb are queued on one of K streams and the call returns; the K-th call
 * after it waits for it.  h_rgb holds the frame after rt_sync(ctx, NULL), and should be page-locked
 * (rt_pinned_alloc) for the copy to run beside the next frames' kernels.  The driver loop of the reference
 * (raytracer.c:47-59) runs about twice as fast this way. */
int rt_render_frame_host(rt_ctx* ctx, const rt_camera* camera, const rt_light* light,
                         const rt_transform* transforms, int32_t n_transforms,
                         int32_t width, int32_t height, uint32_t flags,
                         uint8_t* h_rgb, int32_t* h_hit_id, float* h_hit_t, uint8_t* h_shadow);
/* Page-locked host memory for the host-buffer entry points: copies into it run at bus speed and do not
 * fault pages in (Scene.frames of the host API library are allocated this way when a GPU is present).
 * RT_ERR_NO_DEVICE without a GPU -- fall back to malloc. */
int rt_pinned_alloc(size_t bytes, void** out);
int rt_pinned_free(void* p);

/* ---- encoder-side upscale -------------------------------------------------------------- */
/* Output size save_scene derives from the scene (scene.c:164-165): (int)(size / scale_factor + 0.5f). */
int rt_scaled_size(int32_t width, int32_t height, float scale_factor, int32_t* scaled_width, int32_t* scaled_height);

/* The bicubic upscale loop of save_scene (scene.c:188-202, bicubic_interpolate utils/image.c:25-78) for
 * n_frames frames: RGB8 frame f at (uint8_t*)d_rgb + f*frame_stride (bytes; the layout rt_render_frames
 * writes) becomes the ARGB32 image WebPPicture.argb holds, 0xFF000000 | r<<16 | g<<8 | b, row-major
 * scaled_width x scaled_height, at (uint32_t*)d_argb + f*out_stride (out_stride in PIXELS, >=
 * scaled_width*scaled_height).  scaled sizes must be >= 2.  Asynchronous on `stream`. */
int rt_upscale_frames(rt_ctx* ctx, const void* d_rgb, size_t frame_stride, int32_t n_frames,
                      int32_t width, int32_t height, int32_t scaled_width, int32_t scaled_height,
                      void* d_argb, size_t out_stride, void* stream);
/* Same for one frame in host memory: copies in, upscales, waits, copies out. */
int rt_upscale_frame_host(rt_ctx* ctx, const uint8_t* h_rgb, int32_t width, int32_t height,
                          int32_t scaled_width, int32_t scaled_height, uint32_t* h_argb);

int rt_sync(rt_ctx* ctx, void* stream);     /* waits for all launches on stream; NULL: the context's stream and frames in flight */
/* GPU time of the most recent rt_render_frame launch, from HIP events recorded on its stream
 * (waits for that launch). */
int rt_last_kernel_ms(rt_ctx* ctx, float* ms);
/* Counters of the most recent RT_FLAG_COUNT_WORK launch (waits for it). */
int rt_last_counters(rt_ctx* ctx, rt_counters* out);

/* Tracing aid (no counterpart in the reference): while a buffer is set, every launch records for each
 * 8x8 pixel tile, in tile order (row-major over ceil(width/8) x ceil(height/8) tiles), two uint64:
 * the device's constant 100 MHz clock when a wave started and finished that tile.  d_tile_clock
 * must hold 16 bytes per tile of the frames rendered; NULL switches tracing off. */
int rt_set_trace_buffer(rt_ctx* ctx, void* d_tile_clock);

/* Tunables, by name: "variant" (kernel variant id of rt_render_frame, 0 <= id < "n_variants": 0 = one lane per pixel, 3 = the
 * same with cooperative tiles where rays are long, the default; the other ids are experiments that lost against these two and
 * are only in builds made with EXPERIMENTAL=1 -- read-only option "experimental_variants" says whether -- otherwise they run variant 0), "block"
 * (threads per workgroup: 64, 128 or 256), "timing" (record HIP events around each launch, default 1),
 * "coop" (variant 3: 0 off, 1 cooperative tiles where primary or shadow rays can meet a big mesh,
 * 2 primary rays only), "budget" (variant 4: node visits one lane spends on a pixel before the pixel is
 * handed to the cooperative pass), "resident_budget" (variants 5..9: bytes of small meshes staged in LDS),
 * "batch_chunk" (rt_render_frames: consecutive tiles of one frame before the same tiles of the next
 * frame; 0 = four tile rows), "xcd_swizzle" (rt_render_frames: 1 = each XCD gets runs of eight neighbouring
 * tiles, default; 0 = every eighth tile), "device_build_min" (see rt_mesh_upload_soup), "regroup" (1, default: a mesh uploaded through the host
 * packer gets a second tree over the reference's own leaves, split by surface area, which rays enter instead of the
 * reference's mean-split tree, and, where the mesh is made of long thin triangles along the diagonals of a coordinate plane, a
 * further one whose boxes live in the frame sheared by 45 degrees in that plane; the leaves whose triangles get tested are
 * fewer, the triangles accepted, and every output, are the same; 0 = the reference's tree only;
 * applies to meshes uploaded afterwards), "sampled_split" (rt_render_frame_sampled with four samples or
 * more: the samples of a tile are traced by four waves and added up in order instead of one wave looping
 * over them -- 0 never, 1 when the launch has fewer than 100 000 tiles (default), 2 always),
 * "frames_in_flight" (1..8, default 1: see rt_render_frame_host), "keep_hierarchy" (rt_render_frame_instanced:
 * 1, default = the hierarchy over the instances is kept between calls and patched while few instances change
 * their place; 0 = built anew for every call), "fast" (0, default = the exact build of the kernels: every float
 * operation of the reference once, in its order, no fused multiply-adds, correctly rounded division and square root --
 * all outputs bit-identical to the reference; 1 = the fast build for rt_render_frame, rt_render_frame_strips,
 * rt_render_frames and rt_render_frame_instanced: contraction on, hardware reciprocal / reciprocal square root, the slab test
 * as one fused multiply-add per bound, and closed meshes walked through a tree over their triangles themselves (leaves of up
 * to eight neighbouring triangles in tight boxes, built beside the other trees at upload when option "regroup" is on) instead
 * of the reference's leaves.  Not bit-exact: hit ids equal on at least 99.99 % of camera rays and 8-bit colours within 1 on at
 * least 99.9 % of pixels against the exact build, which the GPU tests assert; measured 1.2 times as fast on the animation (a mesh with a tree in a sheared frame keeps that tree),
 * 1.5 times on the instanced stress scene.  Calls with RT_FLAG_VERIFY_SKIPS or RT_FLAG_REFERENCE_ORDER run the exact build
 * whatever this option says; rt_render_frame_sampled always does), "instanced_wave" (rt_render_frame_instanced: 0, default = one lane per pixel walks
 * hierarchy, instance and triangles as nested loops; 1..8 = every lane is a state machine and the wave executes the step most of
 * its lanes wait for, lanes taking new pixels from a strip of that many tiles -- more lanes busy per instruction, more
 * instructions: measured slower, kept for comparison; it culls by distance inside every instance, so a list that places a mesh
 * with a triangle of |e1| |e2| >= 0.25, which is walked without that cull, is rendered by the default kernel all the same).
 * Every setting except "fast" produces the same image.  Read-only: "n_variants", "default_variant".
 * Unknown names give RT_ERR_INVALID. */
int rt_set_option(rt_ctx* ctx, const char* name, int value);
int rt_get_option(const rt_ctx* ctx, const char* name, int* value);

#ifdef __cplusplus
}
#endif
#endif
