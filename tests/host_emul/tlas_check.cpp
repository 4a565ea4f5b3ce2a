// tlas_check.cpp -- test program (not product code): drives the instance hierarchy of rt_pack.h through
// random sequences of changes and checks after every step that the tree is a valid hierarchy for the
// current list: every non-empty instance hangs in exactly one reachable leaf, no other instance is
// reachable, and every box contains the boxes below it.  Built with -fsanitize=address,undefined by
// tests/test_instanced.py.
#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../../raytracer_amd/csrc/rt_pack.h"

using namespace rt;

namespace {

int g_errors = 0;
void fail(const char* what, int step) { if (g_errors++ < 10) fprintf(stderr, "step %d: %s\n", step, what); }

bool inside(const float* lo, const float* hi, const float* plo, const float* phi) {
    for (int i = 0; i < 3; i++) if (lo[i] < plo[i] || hi[i] > phi[i]) return false;
    return true;
}

// Walks the subtree under `ref`, whose box in the parent is [plo, phi]; counts how often each instance is reached.
void walk(const Tlas& T, uint32_t ref, const float* plo, const float* phi, const std::vector<InstanceBox>& box_of, std::vector<int>& seen,
          int depth, int& deepest, int step) {
    if (depth > deepest) deepest = depth;
    if (depth > 64) { fail("tree deeper than 64", step); return; }
    if (ref & kLeafBit) {
        const uint32_t first = ref & 0x00FFFFFFu, count = (ref >> 24) & 0x7Fu;
        for (uint32_t k = 0; k < count; k++) {
            if (first + k >= T.order.size()) { fail("leaf points past the order list", step); return; }
            const uint32_t id = T.order[first + k];
            if (id >= seen.size()) { fail("instance id out of range", step); return; }
            seen[id]++;
            if (!inside(box_of[id].lo, box_of[id].hi, plo, phi)) fail("instance box not inside its leaf's box", step);
        }
        return;
    }
    if (ref >= T.pairs.size()) { fail("node index out of range", step); return; }
    const PairNode& p = T.pairs[ref];
    if (!(p.lref & kLeafBit) || ((p.lref >> 24) & 0x7Fu)) if (!inside(p.lmin, p.lmax, plo, phi)) fail("left box not inside parent", step);
    if (!(p.rref & kLeafBit) || ((p.rref >> 24) & 0x7Fu)) if (!inside(p.rmin, p.rmax, plo, phi)) fail("right box not inside parent", step);
    walk(T, p.lref, p.lmin, p.lmax, box_of, seen, depth + 1, deepest, step);
    walk(T, p.rref, p.rmin, p.rmax, box_of, seen, depth + 1, deepest, step);
}

}  // namespace

int main(int argc, char** argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int steps = argc > 2 ? atoi(argv[2]) : 300;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> pos(-50.0f, 50.0f), ext(0.1f, 3.0f), unit01(0.0f, 1.0f);
    StandingTlas standing;
    size_t n = 200;
    std::vector<InstanceBox> box_of(n);
    std::vector<uint8_t> present(n, 1);
    auto fresh = [&](InstanceBox& b) { for (int i = 0; i < 3; i++) { b.lo[i] = pos(rng); b.hi[i] = b.lo[i] + ext(rng); } };
    for (auto& b : box_of) fresh(b);
    int reused = 0;
    for (int step = 0; step < steps; step++) {
        const float what = unit01(rng);
        if (what < 0.05f) {                                   // another list length
            n = 1 + (size_t)(unit01(rng) * 400.0f);
            box_of.resize(n); present.assign(n, 1);
            for (auto& b : box_of) fresh(b);
        } else if (what < 0.15f) {                            // many movers
            for (size_t i = 0; i < n; i += 2) fresh(box_of[i]);
        } else if (what < 0.9f) {                             // a few movers, emptied and refilled instances
            const int k = 1 + (int)(unit01(rng) * 6.0f);
            for (int j = 0; j < k; j++) {
                const size_t i = (size_t)(unit01(rng) * (float)n) % n;
                if (unit01(rng) < 0.2f) present[i] = !present[i];
                else fresh(box_of[i]);
            }
        }                                                     // else: nothing changes
        std::vector<InstanceBox> boxes;
        std::vector<uint32_t> ids;
        for (size_t i = 0; i < n; i++) if (present[i]) { boxes.push_back(box_of[i]); ids.push_back((uint32_t)i); }
        Tlas T;
        reused += standing.update(boxes, ids, n, T) ? 1 : 0;
        std::vector<int> seen(n, 0);
        int deepest = 0;
        if (!ids.empty()) walk(T, T.root_ref, T.root_min, T.root_max, box_of, seen, 0, deepest, step);
        for (size_t i = 0; i < n; i++) if (seen[i] != (present[i] ? 1 : 0)) { fail("an instance is reached the wrong number of times", step); break; }
        if (deepest > T.depth) fail("tree deeper than it says", step);
        if (T.depth + 2 > kTlasStackDepth) fail("tree deeper than the walk's stack", step);
    }
    printf("{\"steps\": %d, \"reused\": %d, \"errors\": %d}\n", steps, reused, g_errors);
    return g_errors ? 1 : 0;
}
