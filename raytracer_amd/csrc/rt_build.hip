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
            // The reference folds every corner into a box that starts at +-1e30 with fminf / fmaxf (aabb.c:3-23): a NaN never
            // gets in, nor does anything beyond the starting value.  Such a bound takes no part here either.
            if (!(pick(lo, a) <= 1e30f)) k[a] = ~0ull;
            if (!(pick(hi, a) >= -1e30f)) k[3 + a] = 0ull;
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
        // no bound took part (all NaN or beyond the starting value): the box keeps what it started with (aabb.c:3-8)
        nd.bmin[a] = kmin == ~0ull ? 1e30f : pick(box[2 * (size_t)perm[pmin]], a);
        nd.bmax[a] = kmax == 0ull ? -1e30f : pick(box[2 * (size_t)perm[pmax] + 1], a);
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

// ---- packing for the traversal kernels (what Packer in rt_pack.h does on the host) ---------------------
// PairNodes are emitted in creation order of the internal nodes (node, left subtree, right subtree).
__global__ void k_internal_sizes(const BfsNode* nodes, int32_t first, int32_t count, int32_t* isz, int32_t* max_leaf) {
    const int32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const BfsNode& nd = nodes[first + i];
    if (nd.left >= 0) isz[first + i] = 1 + isz[nd.left] + isz[nd.right];
    else { isz[first + i] = 0; atomicMax(max_leaf, nd.count); }
}

__global__ void k_internal_number(const BfsNode* nodes, int32_t first, int32_t count, const int32_t* isz, int32_t* inum) {
    const int32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const int32_t n = first + i;
    const BfsNode& nd = nodes[n];
    if (nd.left < 0) return;
    const int32_t me = n == 0 ? 0 : inum[n];
    if (n == 0) inum[0] = 0;
    inum[nd.left] = me + 1;                              // only read if that child is internal
    inum[nd.right] = me + 1 + isz[nd.left];
}

__global__ void k_pairs(const BfsNode* nodes, int32_t n_nodes, const int32_t* inum, PairNode* pairs) {
    const int32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_nodes) return;
    const BfsNode nd = nodes[n];
    if (nd.left < 0) return;
    const BfsNode l = nodes[nd.left], r = nodes[nd.right];
    PairNode p;
    for (int a = 0; a < 3; a++) { p.lmin[a] = l.bmin[a]; p.lmax[a] = l.bmax[a]; p.rmin[a] = r.bmin[a]; p.rmax[a] = r.bmax[a]; }
    p.lref = l.left >= 0 ? (uint32_t)inum[nd.left] : leaf_ref((uint32_t)l.start, (uint32_t)l.count);
    p.rref = r.left >= 0 ? (uint32_t)inum[nd.right] : leaf_ref((uint32_t)r.start, (uint32_t)r.count);
    p.pad[0] = p.pad[1] = 0;
    pairs[n == 0 ? 0 : inum[n]] = p;
}

// Tie order of the reference as a per-triangle priority: a later leaf beats an earlier one, inside a leaf
// the lower index wins.  Leaves tile the triangle array in creation order, so the running count before a
// leaf is its start.
__global__ void k_leaf_prio(const BfsNode* nodes, int32_t n_nodes, uint32_t* prio) {
    const int32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_nodes) return;
    const BfsNode nd = nodes[n];
    if (nd.left >= 0) return;
    for (int32_t i = 0; i < nd.count; i++) prio[nd.start + i] = (uint32_t)(nd.start + (nd.count - i));
}

__global__ void k_records(const rt_triangle* tris, const uint32_t* prio, int32_t n, TriRec* tr, AttrRec* at) {
    const int32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const rt_triangle s = tris[i];
    TriRec t;
    t.v0[0] = s.v0.x; t.v0[1] = s.v0.y; t.v0[2] = s.v0.z; t.prio = prio[i];
    t.e1[0] = s.v1.x - s.v0.x; t.e1[1] = s.v1.y - s.v0.y; t.e1[2] = s.v1.z - s.v0.z; t.pad1 = __uint_as_float((uint32_t)i);      // ray.c:45-46; the triangle's id
    t.e2[0] = s.v2.x - s.v0.x; t.e2[1] = s.v2.y - s.v0.y; t.e2[2] = s.v2.z - s.v0.z; t.pad2 = 0.0f;
    tr[i] = t;
    AttrRec a;
    a.t0[0] = s.t0.u; a.t0[1] = s.t0.v; a.t1[0] = s.t1.u; a.t1[1] = s.t1.v; a.t2[0] = s.t2.u; a.t2[1] = s.t2.v;
    a.n0[0] = s.n0.x; a.n0[1] = s.n0.y; a.n0[2] = s.n0.z;
    a.n1[0] = s.n1.x; a.n1[1] = s.n1.y; a.n1[2] = s.n1.z;
    a.n2[0] = s.n2.x; a.n2[1] = s.n2.y; a.n2[2] = s.n2.z;
    a.pad = 0.0f;
    at[i] = a;
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

struct DeviceBvh::Impl {
    std::string err;
    Arena A;
    rt_triangle* d_in = nullptr; rt_triangle* d_out = nullptr;
    BfsNode* nodes = nullptr; int32_t* size = nullptr; int32_t* number = nullptr; rt_flat_node* d_flat = nullptr;
    int32_t* perm = nullptr;
    std::vector<int32_t> level_first;      // first breadth-first node id of every level, then the node count
    int32_t n = 0, total_nodes = 0;
    bool flat_done = false;
    Impl() { A.err = &err; }
};

DeviceBvh::~DeviceBvh() { delete impl; }

int DeviceBvh::build(const rt_triangle* h_tris, int32_t n, hipStream_t stream, std::string& err) {
    delete impl;
    impl = new Impl;
    Impl& I = *impl;
    Arena& A = I.A;
    stats = BuildStats();
    I.n = n;
    const size_t N = (size_t)n;
    rt_triangle* d_in = I.d_in = A.get<rt_triangle>(N);
    I.d_out = A.get<rt_triangle>(N);
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
    BfsNode* nodes = I.nodes = A.get<BfsNode>(2 * N);
    I.size = A.get<int32_t>(2 * N);
    I.number = A.get<int32_t>(2 * N);
    I.d_flat = A.get<rt_flat_node>(2 * N);
    LevelCounters* d_cnt = A.get<LevelCounters>(1);
    size_t scan_bytes = 0;
    if (A.ok) BUILD_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, less, lsum, 0u, N + 1, rocprim::plus<uint32_t>(), stream));
    uint8_t* scan_tmp = A.get<uint8_t>(scan_bytes);
    if (!A.ok) { err = I.err; return RT_ERR_DEVICE; }
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

    std::vector<int32_t>& level_first = I.level_first;
    level_first.assign(1, 0);
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
    if (level_first.back() == total_nodes) level_first.pop_back();            // the last round created nothing
    level_first.push_back(total_nodes);
    I.total_nodes = total_nodes;
    I.perm = perm[cur];
    hipLaunchKernelGGL(k_gather, dim3(blocks((int64_t)n * 6)), dim3(kBlock), 0, stream, (const float4*)d_in, I.perm, n, (float4*)I.d_out);
    launches += 1;
    BUILD_HIP(hipEventRecord(ev1, stream));
    BUILD_HIP(hipStreamSynchronize(stream));
    BUILD_HIP(hipGetLastError());
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) stats.device_ms = ms;
    stats.levels = (int)level_first.size() - 1; stats.launches = launches; stats.nodes = total_nodes;
    return RT_OK;
}

// Flat node array in creation order + the triangles in post-build order, to the host.
int DeviceBvh::download(rt_triangle* h_tris, rt_flat_node* h_nodes, int32_t cap, int32_t* n_nodes, hipStream_t stream, std::string& err) {
    if (!impl) { err = "no build"; return RT_ERR_INVALID; }
    Impl& I = *impl;
    if (I.total_nodes > cap) { err = "node array too small"; return RT_ERR_CAPACITY; }
    const int n_levels = (int)I.level_first.size() - 1;
    for (int l = n_levels - 1; l >= 0; l--) {
        const int32_t c = I.level_first[l + 1] - I.level_first[l];
        hipLaunchKernelGGL(k_sizes, dim3(blocks(c)), dim3(kBlock), 0, stream, I.nodes, I.level_first[l], c, I.size);
    }
    for (int l = 0; l < n_levels; l++) {
        const int32_t c = I.level_first[l + 1] - I.level_first[l];
        hipLaunchKernelGGL(k_number, dim3(blocks(c)), dim3(kBlock), 0, stream, I.nodes, I.level_first[l], c, I.size, I.number);
    }
    hipLaunchKernelGGL(k_emit, dim3(blocks(I.total_nodes)), dim3(kBlock), 0, stream, I.nodes, I.total_nodes, I.number, I.d_flat);
    stats.launches += 2 * n_levels + 1;
    BUILD_HIP(hipMemcpyAsync(h_nodes, I.d_flat, (size_t)I.total_nodes * sizeof(rt_flat_node), hipMemcpyDeviceToHost, stream));
    if (h_tris) BUILD_HIP(hipMemcpyAsync(h_tris, I.d_out, (size_t)I.n * sizeof(rt_triangle), hipMemcpyDeviceToHost, stream));
    BUILD_HIP(hipStreamSynchronize(stream));
    BUILD_HIP(hipGetLastError());
    *n_nodes = I.total_nodes;
    return RT_OK;
}

int DeviceBvh::triangles_to_host(rt_triangle* h_tris, hipStream_t stream, std::string& err) {
    if (!impl) { err = "no build"; return RT_ERR_INVALID; }
    BUILD_HIP(hipMemcpyAsync(h_tris, impl->d_out, (size_t)impl->n * sizeof(rt_triangle), hipMemcpyDeviceToHost, stream));
    BUILD_HIP(hipStreamSynchronize(stream));
    return RT_OK;
}

// The records the traversal kernels read, made where the tree is.  d_tris / d_attrs are new device arrays
// owned by the caller; pairs come back to the host (the scene's combined node array is assembled there).
// RT_ERR_UNSUPPORTED (nothing allocated) when a leaf has more triangles than one leaf reference holds or
// the tree is deeper than the traversal stack: the host packer handles or reports those.
int DeviceBvh::pack(PackedOnDevice& out, hipStream_t stream, std::string& err) {
    if (!impl) { err = "no build"; return RT_ERR_INVALID; }
    Impl& I = *impl;
    const int n_levels = (int)I.level_first.size() - 1;
    if (n_levels - 1 > RT_MAX_TREE_DEPTH || n_levels + 1 > kStackDepth) { err = "BVH deeper than the traversal stack"; return RT_ERR_UNSUPPORTED; }
    Arena tmp;
    tmp.err = &err;
    int32_t* isz = tmp.get<int32_t>((size_t)I.total_nodes);
    int32_t* inum = tmp.get<int32_t>((size_t)I.total_nodes);
    uint32_t* prio = tmp.get<uint32_t>((size_t)I.n);
    int32_t* max_leaf = tmp.get<int32_t>(1);
    if (!tmp.ok) return RT_ERR_DEVICE;
    BUILD_HIP(hipMemsetAsync(max_leaf, 0, sizeof(int32_t), stream));
    for (int l = n_levels - 1; l >= 0; l--) {
        const int32_t c = I.level_first[l + 1] - I.level_first[l];
        hipLaunchKernelGGL(k_internal_sizes, dim3(blocks(c)), dim3(kBlock), 0, stream, I.nodes, I.level_first[l], c, isz, max_leaf);
    }
    int32_t h_info[2] = {0, 0};
    BUILD_HIP(hipMemcpyAsync(&h_info[0], max_leaf, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    BUILD_HIP(hipMemcpyAsync(&h_info[1], isz, sizeof(int32_t), hipMemcpyDeviceToHost, stream));      // internal nodes of the whole tree
    BUILD_HIP(hipStreamSynchronize(stream));
    if (h_info[0] > kMaxLeafTris) { err = "a leaf holds more triangles than one leaf reference"; return RT_ERR_UNSUPPORTED; }
    const int32_t n_pairs = h_info[1];
    for (int l = 0; l < n_levels; l++) {
        const int32_t c = I.level_first[l + 1] - I.level_first[l];
        hipLaunchKernelGGL(k_internal_number, dim3(blocks(c)), dim3(kBlock), 0, stream, I.nodes, I.level_first[l], c, isz, inum);
    }
    PairNode* d_pairs = tmp.get<PairNode>((size_t)n_pairs);
    TriRec* d_tris = nullptr; AttrRec* d_attrs = nullptr;
    if (!tmp.ok) return RT_ERR_DEVICE;
    if (hipMalloc((void**)&d_tris, (size_t)I.n * sizeof(TriRec)) != hipSuccess || hipMalloc((void**)&d_attrs, (size_t)I.n * sizeof(AttrRec)) != hipSuccess) {
        (void)hipGetLastError();
        if (d_tris) (void)hipFree(d_tris);
        err = "hipMalloc of the packed mesh failed";
        return RT_ERR_DEVICE;
    }
    hipLaunchKernelGGL(k_pairs, dim3(blocks(I.total_nodes)), dim3(kBlock), 0, stream, I.nodes, I.total_nodes, inum, d_pairs);
    hipLaunchKernelGGL(k_leaf_prio, dim3(blocks(I.total_nodes)), dim3(kBlock), 0, stream, I.nodes, I.total_nodes, prio);
    hipLaunchKernelGGL(k_records, dim3(blocks(I.n)), dim3(kBlock), 0, stream, I.d_out, prio, I.n, d_tris, d_attrs);
    out.pairs.resize((size_t)n_pairs);
    BfsNode root;
    hipError_t e = hipMemcpyAsync(&root, I.nodes, sizeof(BfsNode), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && n_pairs > 0) e = hipMemcpyAsync(out.pairs.data(), d_pairs, (size_t)n_pairs * sizeof(PairNode), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipFree(d_tris); (void)hipFree(d_attrs);
        err = std::string("device packing: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return RT_ERR_DEVICE;
    }
    out.d_tris = d_tris; out.d_attrs = d_attrs;
    memcpy(out.root_min, root.bmin, 12); memcpy(out.root_max, root.bmax, 12);
    out.root_ref = root.left >= 0 ? 0u : leaf_ref(0u, (uint32_t)root.count);
    out.depth = n_levels - 1;
    return RT_OK;
}

int build_bvh_on_device(rt_triangle* h_tris, int32_t n, rt_flat_node* h_nodes, int32_t cap, int32_t* n_nodes, hipStream_t stream,
                        std::string& err, BuildStats* stats) {
    if (stats) *stats = BuildStats();
    DeviceBvh b;
    int rc = b.build(h_tris, n, stream, err);
    if (rc == RT_OK) rc = b.download(h_tris, h_nodes, cap, n_nodes, stream, err);
    if (stats) *stats = b.stats;
    return rc;
}

}  // namespace rt
