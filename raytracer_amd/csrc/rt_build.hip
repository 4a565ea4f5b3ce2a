// rt_build.hip -- the reference's BVH build (accel/bvh.c:4-59) on the GPU, bit for bit.
//
// The reference builds depth first: bounds of the node's triangles, longest axis, the mean of the
// centroids along it (a float sum in array order), a swap-to-front partition around that mean,
// then the two halves.  Nodes of one depth touch disjoint ranges of the triangle array, so the same
// tree comes out when all nodes of a depth are processed together.  Per level:
//
//   bounds      min/max are exact in any order; only the sign of a zero bound depends on order (the C
//               library's fminf/fmaxf return the later operand when +0 meets -0), so the reduction
//               runs on 64-bit keys (ordered value, position) and the winner's own bits are stored.
//   mean        the float sum has one rounding per addition in array order and cannot be reassociated:
//               one wavefront per node loads 64 keys at a time and adds them one lane after the other.
//               This is the serial part of the build (about 10 cycles per triangle of the largest node).
//   partition   "for i: if key[i] < split: swap(a[i], a[mid++])" keeps the smaller keys in order and
//               rotates the others through a queue (the front one moves behind the last each time a
//               smaller key arrives).  Written as a log of queue insertions, slot k of the log is
//               either a new element or a copy of an earlier slot; the final order is the tail of the
//               log.  The copies are resolved for all slots at once by pointer doubling.
//
// Nodes are numbered breadth first while building and renumbered to the reference's creation order
// (node, left subtree, right subtree) at the end, from subtree sizes.
#include <hip/hip_runtime.h>
#include <string.h>

#include <rocprim/device/device_scan.hpp>

#include <string>
#include <vector>

#include "rt_build.h"
#include "rt_exact.h"

namespace rt {
namespace {

struct Seg { int32_t start, count, node; };

struct BfsNode {
    float bmin[3], bmax[3];
    int32_t start, count, left, right;
};

struct LevelCounters { int32_t n_next; int32_t max_count; int32_t n_nodes; int32_t pad; };

constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t ordered_bits(float v) {
    if (v == 0.0f) v = 0.0f;                                   // one key for both zeros
    const uint32_t u = __float_as_uint(v);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}

// ---- once per build --------------------------------------------------------------------------
// keys = the three coordinates of ((v0+v1)+v2) * (1.0f/3.0f) (bvh.c:30); per-triangle bounds folded in the
// order the reference expands a box (v0, v1, v2; aabb.c:10-23).  One 16-byte and one 32-byte record per
// triangle: later passes fetch them through the permutation, a sector or two per triangle.
__global__ void k_prepare(const rt_triangle* tris, int32_t n, float4* keys, float4* box, int32_t* perm, int32_t* seg_of) {
    const int32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    const float* v = &tris[t].v0.x;
    const float third = 1.0f / 3.0f;
    float k[3], lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
        const float p0 = v[a], p1 = v[3 + a], p2 = v[6 + a];
        k[a] = ((p0 + p1) + p2) * third;
        lo[a] = min_later(min_later(p0, p1), p2);
        hi[a] = max_later(max_later(p0, p1), p2);
    }
    keys[t] = make_float4(k[0], k[1], k[2], 0.0f);
    box[2 * (size_t)t] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    box[2 * (size_t)t + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    perm[t] = t;
    seg_of[t] = 0;
}

__device__ __forceinline__ float pick(const float4& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// ---- per level ---------------------------------------------------------------------------------
// While a level has few nodes, thousands of workgroups reduce into the same node: each node then has
// kSpread copies of its six keys (a workgroup picks one by its index) and k_node_setup merges them.
constexpr int kSpread = 32;
constexpr int kSpreadSegs = 1024;

__global__ void k_init_keys(unsigned long long* red, int32_t nseg, int spread) {
    const int32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < nseg * spread * 6) red[i] = (i % 6) < 3 ? ~0ull : 0ull;
}

// Segmented reduction of the six bound keys.  Positions of one node are contiguous, so inside a wave
// each node is a run of lanes: shuffle-down doubling leaves a run's result in its first lane.
__global__ void k_bounds(const int32_t* seg_of, const int32_t* perm, const Seg* segs, const float4* box, int32_t n,
                         unsigned long long* red, int spread) {
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t s = p < n ? seg_of[p] : -1;
    unsigned long long k[6];
    if (s >= 0) {
        const int32_t t = perm[p];
        const float4 lo = box[2 * (size_t)t], hi = box[2 * (size_t)t + 1];
        for (int a = 0; a < 3; a++) {
            k[a] = ((unsigned long long)ordered_bits(pick(lo, a)) << 32) | (uint32_t)~(uint32_t)p;       // ties: later position is smaller
            k[3 + a] = ((unsigned long long)ordered_bits(pick(hi, a)) << 32) | (uint32_t)p;              // ties: later position is larger
        }
    } else {
        for (int a = 0; a < 3; a++) { k[a] = ~0ull; k[3 + a] = 0ull; }
    }
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t so = __shfl_down(s, d, 64);
        const bool take = lane + d < 64 && so == s;
        for (int a = 0; a < 6; a++) {
            const unsigned long long o = __shfl_down(k[a], d, 64);
            if (take) k[a] = a < 3 ? (o < k[a] ? o : k[a]) : (o > k[a] ? o : k[a]);
        }
    }
    // a workgroup that lies inside one node sends one set of keys, not four
    __shared__ unsigned long long part[4][6];
    __shared__ int32_t part_seg[4];
    const int32_t s0 = __shfl(s, 0, 64);
    const bool wave_uniform = __all(s == s0) && s0 >= 0;
    if (lane == 0) {
        part_seg[wave] = wave_uniform ? s0 : -2 - wave;
        for (int a = 0; a < 6; a++) part[wave][a] = k[a];
    }
    __syncthreads();
    const bool block_uniform = part_seg[0] == part_seg[1] && part_seg[1] == part_seg[2] && part_seg[2] == part_seg[3];
    const int32_t prev = __shfl_up(s, 1, 64);
    bool head = s >= 0 && (lane == 0 || prev != s);
    if (block_uniform) {
        head = threadIdx.x == 0;
        if (head)
            for (int w = 1; w < 4; w++)
                for (int a = 0; a < 6; a++) k[a] = a < 3 ? (part[w][a] < k[a] ? part[w][a] : k[a]) : (part[w][a] > k[a] ? part[w][a] : k[a]);
    }
    if (!head) return;
    const Seg g = segs[s];
    const int32_t wave_base = p - lane;
    const bool whole = spread == 1 && g.start >= wave_base && g.start + g.count <= wave_base + 64;      // nobody else contributes
    unsigned long long* r = red + ((size_t)s * spread + (blockIdx.x % spread)) * 6;
    for (int a = 0; a < 6; a++) {
        if (whole) r[a] = k[a];
        else if (a < 3) atomicMin(&r[a], k[a]);
        else atomicMax(&r[a], k[a]);
    }
}

// Writes the node and picks the axis (bvh.c:21-24); axis -1 = not split (bvh.c:19).
__global__ void k_node_setup(const Seg* segs, int32_t nseg, const unsigned long long* red, int spread, const int32_t* perm, const float4* box,
                             BfsNode* nodes, int32_t* axis) {
    const int32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= nseg) return;
    const Seg g = segs[s];
    BfsNode nd;
    for (int a = 0; a < 3; a++) {
        unsigned long long kmin = ~0ull, kmax = 0ull;
        for (int c = 0; c < spread; c++) {
            const unsigned long long lo = red[((size_t)s * spread + c) * 6 + a], hi = red[((size_t)s * spread + c) * 6 + 3 + a];
            kmin = lo < kmin ? lo : kmin; kmax = hi > kmax ? hi : kmax;
        }
        const uint32_t pmin = ~(uint32_t)(kmin & 0xFFFFFFFFull);
        const uint32_t pmax = (uint32_t)(kmax & 0xFFFFFFFFull);
        nd.bmin[a] = pick(box[2 * (size_t)perm[pmin]], a);
        nd.bmax[a] = pick(box[2 * (size_t)perm[pmax] + 1], a);
    }
    nd.start = g.start; nd.count = g.count; nd.left = nd.right = -1;
    nodes[g.node] = nd;
    int ax = -1;
    if (g.count > 4) {
        const float ex = nd.bmax[0] - nd.bmin[0], ey = nd.bmax[1] - nd.bmin[1], ez = nd.bmax[2] - nd.bmin[2];
        ax = 0;
        if (ey > ex) ax = 1;
        if (ez > ex && ez > ey) ax = 2;
    }
    axis[s] = ax;
}

// The keys of this level in array order: what the mean adds up and the partition compares.
__global__ void k_level_keys(const int32_t* seg_of, const int32_t* perm, const int32_t* axis, const float4* keys, int32_t n, float* kk) {
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const int32_t s = seg_of[p];
    float v = 0.0f;
    if (s >= 0) {
        const int ax = axis[s];
        if (ax >= 0) v = pick(keys[perm[p]], ax);
    }
    kk[p] = v;
}

// split = (((0 + k0) + k1) + ...) / count (bvh.c:27-33): one wave per node, one addition after the other.
// Each lane holds one key of a 64-key chunk.  The chain for the 16 keys of a row of lanes runs in the
// row's first lane: step k adds the chunk shifted left by k lanes (a DPP move that does not depend on the
// sum, and is the same for all four rows), so the only dependent instructions are the additions
// themselves; after a row the sum is read from its lane and continues in the next row.  About 1.3
// instructions per key.  (Measured alternatives on 10 M keys: reading every key into a scalar register,
// 79 ms; passing the sum from lane to lane inside the add with a wave rotate, which has to wait two
// issue slots for its own result, 58 ms.)  Lanes hold zeros beyond the end of the node: the sum is never
// -0, so adding +0 leaves its bits alone.  Keys are loaded kTile chunks ahead, which covers the memory
// latency of the next tile with the additions of this one.
constexpr int kTile = 16;

template <int K> __device__ __forceinline__ float shifted(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + K /* row_shl:K */, 0xF, 0xF, true));
}

__device__ __forceinline__ float add_chunk(float sum, float v) {
    const float v1 = shifted<1>(v), v2 = shifted<2>(v), v3 = shifted<3>(v), v4 = shifted<4>(v), v5 = shifted<5>(v), v6 = shifted<6>(v),
                v7 = shifted<7>(v), v8 = shifted<8>(v), v9 = shifted<9>(v), v10 = shifted<10>(v), v11 = shifted<11>(v), v12 = shifted<12>(v),
                v13 = shifted<13>(v), v14 = shifted<14>(v), v15 = shifted<15>(v);
#pragma unroll
    for (int row = 0; row < 4; row++) {
        float s = sum + v;                                         // meaningful in lane 16*row
        s = s + v1; s = s + v2; s = s + v3; s = s + v4; s = s + v5; s = s + v6; s = s + v7; s = s + v8;
        s = s + v9; s = s + v10; s = s + v11; s = s + v12; s = s + v13; s = s + v14; s = s + v15;
        sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 16 * row));
    }
    return sum;
}

__global__ void k_serial_mean(const Seg* segs, int32_t nseg, const int32_t* axis, const float* kk, float* split) {
    const int32_t s = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (s >= nseg) return;
    if (axis[s] < 0) return;
    const Seg g = segs[s];
    const float* k = kk + g.start;
    float sum = 0.0f;
    float cur[kTile], nxt[kTile];
#pragma unroll
    for (int j = 0; j < kTile; j++) { const int32_t i = j * 64 + lane; cur[j] = i < g.count ? k[i] : 0.0f; }
    for (int32_t base = 0; base < g.count; base += 64 * kTile) {
#pragma unroll
        for (int j = 0; j < kTile; j++) { const int32_t i = base + 64 * kTile + j * 64 + lane; nxt[j] = i < g.count ? k[i] : 0.0f; }
#pragma unroll
        for (int j = 0; j < kTile; j++)
            if (base + j * 64 < g.count) sum = add_chunk(sum, cur[j]);      // wave-uniform
#pragma unroll
        for (int j = 0; j < kTile; j++) cur[j] = nxt[j];
    }
    if (lane == 0) split[s] = sum / (float)g.count;
}

__global__ void k_flags(const int32_t* seg_of, const int32_t* axis, const float* split, const float* kk, int32_t n, uint32_t* less) {
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p > n) return;
    uint32_t f = 0;
    if (p < n) {
        const int32_t s = seg_of[p];
        if (s >= 0 && axis[s] >= 0) f = kk[p] < split[s] ? 1u : 0u;                     // bvh.c:42
    }
    less[p] = f;                                                                        // less[n] = 0 closes the scan
}

// Left count, leading run of smaller keys, children (bvh.c:51-55).
__global__ void k_split_setup(const Seg* segs, int32_t nseg, const int32_t* axis, const uint32_t* lsum, BfsNode* nodes, Seg* next, int32_t* child,
                              int32_t* lead, LevelCounters* cnt) {
    const int32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= nseg) return;
    child[s] = -1;
    if (axis[s] < 0) return;
    const Seg g = segs[s];
    const uint32_t base = lsum[g.start];
    const int32_t nl = (int32_t)(lsum[g.start + g.count] - base);
    if (nl <= 0 || nl >= g.count) return;
    int32_t lo = 0, hi = g.count;                       // first `lo` keys are all smaller; not all of the first `hi`
    while (hi - lo > 1) {
        const int32_t mid = lo + (hi - lo) / 2;
        if ((int32_t)(lsum[g.start + mid] - base) == mid) lo = mid; else hi = mid;
    }
    lead[s] = lo;
    const int32_t ci = atomicAdd(&cnt->n_next, 2);
    const int32_t ni = atomicAdd(&cnt->n_nodes, 2);
    next[ci] = Seg{g.start, nl, ni};
    next[ci + 1] = Seg{g.start + nl, g.count - nl, ni + 1};
    nodes[g.node].left = ni; nodes[g.node].right = ni + 1;
    child[s] = ci;
    const int32_t big = nl > g.count - nl ? nl : g.count - nl;
    atomicMax(&cnt->max_count, big);
}

// Slot p of the insertion log: a key that is not smaller is a new element (points at itself); the r-th
// smaller key after the leading run re-inserts the element of slot r.
__global__ void k_log_init(const int32_t* seg_of, const Seg* segs, const int32_t* child, const int32_t* lead, const uint32_t* less,
                           const uint32_t* lsum, int32_t n, int32_t* nxt) {
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    int32_t to = p;
    const int32_t s = seg_of[p];
    if (s >= 0 && child[s] >= 0 && less[p]) {
        const Seg g = segs[s];
        const int32_t z = lead[s];
        if (p - g.start >= z) to = g.start + z + ((int32_t)(lsum[p] - lsum[g.start]) - z);
    }
    nxt[p] = to;
}

__global__ void k_jump(const int32_t* in, int32_t* out, int32_t n) {
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p < n) out[p] = in[in[p]];
}

__global__ void k_scatter(const int32_t* seg_of, const Seg* segs, const int32_t* child, const uint32_t* less, const uint32_t* lsum, const int32_t* nxt,
                          const int32_t* perm, int32_t n, int32_t* perm_out, int32_t* seg_out) {
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const int32_t s = seg_of[p];
    const int32_t c = s >= 0 ? child[s] : -1;
    if (c < 0) { perm_out[p] = perm[p]; seg_out[p] = -1; return; }       // finished: a leaf keeps its order (bvh.c:52)
    const Seg g = segs[s];
    const uint32_t base = lsum[g.start];
    const int32_t nl = (int32_t)(lsum[g.start + g.count] - base);
    if (less[p]) {
        const int32_t to = g.start + (int32_t)(lsum[p] - base);
        perm_out[to] = perm[p]; seg_out[to] = c;
    }
    if (p >= g.start + nl) { perm_out[p] = perm[nxt[p]]; seg_out[p] = c + 1; }
}

// ---- creation-order numbering --------------------------------------------------------------------
__global__ void k_sizes(const BfsNode* nodes, int32_t first, int32_t count, int32_t* size) {
    const int32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const BfsNode& nd = nodes[first + i];
    size[first + i] = nd.left >= 0 ? 1 + size[nd.left] + size[nd.right] : 1;
}

__global__ void k_number(const BfsNode* nodes, int32_t first, int32_t count, const int32_t* size, int32_t* number) {
    const int32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const BfsNode& nd = nodes[first + i];
    if (first + i == 0) number[0] = 0;
    if (nd.left >= 0) {
        const int32_t me = first + i == 0 ? 0 : number[first + i];
        number[nd.left] = me + 1;
        number[nd.right] = me + 1 + size[nd.left];
    }
}

__global__ void k_emit(const BfsNode* nodes, int32_t n_nodes, const int32_t* number, rt_flat_node* out) {
    const int32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    const BfsNode nd = nodes[i];
    rt_flat_node f;
    f.bounds_min = rt_vec3{nd.bmin[0], nd.bmin[1], nd.bmin[2]};
    f.bounds_max = rt_vec3{nd.bmax[0], nd.bmax[1], nd.bmax[2]};
    f.start_idx = nd.start; f.triangle_count = nd.count;
    f.left = nd.left >= 0 ? number[nd.left] : -1;
    f.right = nd.right >= 0 ? number[nd.right] : -1;
    out[number[i]] = f;
}

__global__ void k_gather(const float4* in, const int32_t* perm, int32_t n, float4* out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)n * 6) return;
    const int32_t p = (int32_t)(i / 6), q = (int32_t)(i % 6);
    out[(int64_t)p * 6 + q] = in[(int64_t)perm[p] * 6 + q];
}

// ---- host side -------------------------------------------------------------------------------------
struct Arena {
    std::vector<void*> owned;
    std::string* err;
    bool ok = true;
    template <class T> T* get(size_t count) {
        void* p = nullptr;
        if (!ok) return nullptr;
        const hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) { ok = false; *err = std::string("hipMalloc: ") + hipGetErrorString(e); (void)hipGetLastError(); return nullptr; }
        owned.push_back(p);
        return (T*)p;
    }
    ~Arena() { for (void* p : owned) (void)hipFree(p); }
};

inline int blocks(int64_t items) { return (int)((items + kBlock - 1) / kBlock); }

#define BUILD_HIP(call)                                                                     \
    do {                                                                                    \
        const hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); (void)hipGetLastError(); return RT_ERR_DEVICE; } \
    } while (0)

}  // namespace

int build_bvh_on_device(rt_triangle* h_tris, int32_t n, rt_flat_node* h_nodes, int32_t cap, int32_t* n_nodes, hipStream_t stream,
                        std::string& err, BuildStats* stats) {
    if (stats) *stats = BuildStats();
    Arena A;
    A.err = &err;
    const size_t N = (size_t)n;
    rt_triangle* d_in = A.get<rt_triangle>(N);
    rt_triangle* d_out = A.get<rt_triangle>(N);
    float4* keys = A.get<float4>(N);
    float4* box = A.get<float4>(2 * N);
    float* kk = A.get<float>(N);
    int32_t* perm[2] = {A.get<int32_t>(N), A.get<int32_t>(N)};
    int32_t* seg_of[2] = {A.get<int32_t>(N), A.get<int32_t>(N)};
    int32_t* nxt[2] = {A.get<int32_t>(N), A.get<int32_t>(N)};
    uint32_t* less = A.get<uint32_t>(N + 1);
    uint32_t* lsum = A.get<uint32_t>(N + 1);
    Seg* segs[2] = {A.get<Seg>(N), A.get<Seg>(N)};
    unsigned long long* red = A.get<unsigned long long>(6 * (N > (size_t)kSpreadSegs * kSpread ? N : (size_t)kSpreadSegs * kSpread));
    int32_t* axis = A.get<int32_t>(N);
    float* split = A.get<float>(N);
    int32_t* child = A.get<int32_t>(N);
    int32_t* lead = A.get<int32_t>(N);
    BfsNode* nodes = A.get<BfsNode>(2 * N);
    int32_t* size = A.get<int32_t>(2 * N);
    int32_t* number = A.get<int32_t>(2 * N);
    rt_flat_node* d_flat = A.get<rt_flat_node>(2 * N);
    LevelCounters* d_cnt = A.get<LevelCounters>(1);
    size_t scan_bytes = 0;
    if (A.ok) BUILD_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, less, lsum, 0u, N + 1, rocprim::plus<uint32_t>(), stream));
    uint8_t* scan_tmp = A.get<uint8_t>(scan_bytes);
    if (!A.ok) return RT_ERR_DEVICE;
    LevelCounters* h_cnt = nullptr;
    BUILD_HIP(hipHostMalloc((void**)&h_cnt, sizeof(LevelCounters)));
    struct Pinned { LevelCounters* p; ~Pinned() { (void)hipHostFree(p); } } pinned{h_cnt};

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    BUILD_HIP(hipEventCreate(&ev0));
    BUILD_HIP(hipEventCreate(&ev1));
    struct Events { hipEvent_t a, b; ~Events() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } events{ev0, ev1};

    BUILD_HIP(hipMemcpyAsync(d_in, h_tris, N * sizeof(rt_triangle), hipMemcpyHostToDevice, stream));
    BUILD_HIP(hipEventRecord(ev0, stream));
    hipLaunchKernelGGL(k_prepare, dim3(blocks(n)), dim3(kBlock), 0, stream, d_in, n, keys, box, perm[0], seg_of[0]);
    const Seg root{0, n, 0};
    BUILD_HIP(hipMemcpyAsync(segs[0], &root, sizeof(Seg), hipMemcpyHostToDevice, stream));
    *h_cnt = LevelCounters{0, 0, 1, 0};
    BUILD_HIP(hipMemcpyAsync(d_cnt, h_cnt, sizeof(LevelCounters), hipMemcpyHostToDevice, stream));
    BUILD_HIP(hipStreamSynchronize(stream));             // `root` and *h_cnt are reused below

    std::vector<int32_t> level_first{0};                 // first breadth-first node id of every level
    int32_t nseg = 1, max_count = n, total_nodes = 1;
    int cur = 0, levels = 0, launches = 0;
    while (nseg > 0) {
        if (++levels > kMaxBuildLevels) { err = "tree deeper than the device builder supports"; return RT_ERR_UNSUPPORTED; }
        const int spread = nseg <= kSpreadSegs ? kSpread : 1;
        hipLaunchKernelGGL(k_init_keys, dim3(blocks((int64_t)nseg * spread * 6)), dim3(kBlock), 0, stream, red, nseg, spread);
        hipLaunchKernelGGL(k_bounds, dim3(blocks(n)), dim3(kBlock), 0, stream, seg_of[cur], perm[cur], segs[cur], box, n, red, spread);
        hipLaunchKernelGGL(k_node_setup, dim3(blocks(nseg)), dim3(kBlock), 0, stream, segs[cur], nseg, red, spread, perm[cur], box, nodes, axis);
        launches += 3;
        if (max_count > 4) {
            hipLaunchKernelGGL(k_level_keys, dim3(blocks(n)), dim3(kBlock), 0, stream, seg_of[cur], perm[cur], axis, keys, n, kk);
            hipLaunchKernelGGL(k_serial_mean, dim3(blocks((int64_t)nseg * 64)), dim3(kBlock), 0, stream, segs[cur], nseg, axis, kk, split);
            hipLaunchKernelGGL(k_flags, dim3(blocks((int64_t)n + 1)), dim3(kBlock), 0, stream, seg_of[cur], axis, split, kk, n, less);
            BUILD_HIP(rocprim::exclusive_scan(scan_tmp, scan_bytes, less, lsum, 0u, N + 1, rocprim::plus<uint32_t>(), stream));
            hipLaunchKernelGGL(k_split_setup, dim3(blocks(nseg)), dim3(kBlock), 0, stream, segs[cur], nseg, axis, lsum, nodes, segs[cur ^ 1], child, lead, d_cnt);
            hipLaunchKernelGGL(k_log_init, dim3(blocks(n)), dim3(kBlock), 0, stream, seg_of[cur], segs[cur], child, lead, less, lsum, n, nxt[0]);
            int which = 0;
            for (int64_t reach = 1; reach < max_count; reach *= 2) {          // a chain has fewer links than the node has triangles
                hipLaunchKernelGGL(k_jump, dim3(blocks(n)), dim3(kBlock), 0, stream, nxt[which], nxt[which ^ 1], n);
                which ^= 1;
                launches++;
            }
            hipLaunchKernelGGL(k_scatter, dim3(blocks(n)), dim3(kBlock), 0, stream, seg_of[cur], segs[cur], child, less, lsum, nxt[which], perm[cur], n,
                               perm[cur ^ 1], seg_of[cur ^ 1]);
            launches += 7;
            BUILD_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(LevelCounters), hipMemcpyDeviceToHost, stream));
            BUILD_HIP(hipStreamSynchronize(stream));
            nseg = h_cnt->n_next; max_count = h_cnt->max_count;
            level_first.push_back(total_nodes);
            total_nodes = h_cnt->n_nodes;
            const LevelCounters reset{0, 0, total_nodes, 0};
            *h_cnt = reset;
            BUILD_HIP(hipMemcpyAsync(d_cnt, h_cnt, sizeof(LevelCounters), hipMemcpyHostToDevice, stream));
            BUILD_HIP(hipStreamSynchronize(stream));
            cur ^= 1;
        } else {
            nseg = 0;                                       // every node of this level is a leaf; the order is final
        }
    }
    BUILD_HIP(hipGetLastError());
    if (total_nodes > cap) { err = "node array too small"; return RT_ERR_CAPACITY; }
    if (level_first.back() == total_nodes) level_first.pop_back();            // the last round created nothing
    level_first.push_back(total_nodes);
    const int n_levels = (int)level_first.size() - 1;
    for (int l = n_levels - 1; l >= 0; l--) {
        const int32_t c = level_first[l + 1] - level_first[l];
        hipLaunchKernelGGL(k_sizes, dim3(blocks(c)), dim3(kBlock), 0, stream, nodes, level_first[l], c, size);
    }
    for (int l = 0; l < n_levels; l++) {
        const int32_t c = level_first[l + 1] - level_first[l];
        hipLaunchKernelGGL(k_number, dim3(blocks(c)), dim3(kBlock), 0, stream, nodes, level_first[l], c, size, number);
    }
    hipLaunchKernelGGL(k_emit, dim3(blocks(total_nodes)), dim3(kBlock), 0, stream, nodes, total_nodes, number, d_flat);
    hipLaunchKernelGGL(k_gather, dim3(blocks((int64_t)n * 6)), dim3(kBlock), 0, stream, (const float4*)d_in, perm[cur], n, (float4*)d_out);
    launches += 2 * n_levels + 2;
    BUILD_HIP(hipEventRecord(ev1, stream));
    BUILD_HIP(hipMemcpyAsync(h_nodes, d_flat, (size_t)total_nodes * sizeof(rt_flat_node), hipMemcpyDeviceToHost, stream));
    BUILD_HIP(hipMemcpyAsync(h_tris, d_out, N * sizeof(rt_triangle), hipMemcpyDeviceToHost, stream));
    BUILD_HIP(hipStreamSynchronize(stream));
    BUILD_HIP(hipGetLastError());
    *n_nodes = total_nodes;
    if (stats) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) stats->device_ms = ms;
        stats->levels = n_levels; stats->launches = launches; stats->nodes = total_nodes;
    }
    return RT_OK;
}

}  // namespace rt
