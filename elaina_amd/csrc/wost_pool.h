// wost_pool.h -- the loop shared by every tree query that a whole WAVE answers for all its walkers together (2-D: wost_coop.h;
// 3-D: wost_hip3d.hip).  Device code for gfx950; not part of the C-ABI.
//
// A lane that runs its own descent makes the wave wait for the longest query of its 64 walkers, and a last-level visit --
// exact primitive tests behind per-lane skips -- is executed for the whole wave whenever one lane needs it.  Here the work of
// all walkers of the wave goes through two pools of 4-byte tasks in LDS -- (owner lane << 26 | tree node) and (owner lane << 26
// | leaf slot) -- and every trip of the loop pops up to 64 tasks off ONE pool, whoever owns them, and runs one body on them:
//   node task   measures the four children of the node against its owner's bound (the query's lambda) and hands back either
//               which of the four leaf slots survive (a last-level node) or a key per surviving child (distance bits, child
//               index in the two low bits); the loop pushes slot tasks, or node tasks sorted the farthest first, so that the
//               next trip finds every task's nearest child on top: a 64-wide near-first descent;
//   slot task   tests the primitives of one leaf slot and folds the result into its owner's words with LDS atomics (lambda).
// Every such query is a minimum (closest point / silhouette / first hit), so the order in which candidates are met does not
// matter and the answer is the flat loop's, bit for bit; a bound that is tightened later than a private descent would have
// only costs visits.  The pools are bounded: a trip takes only as many node tasks as leave room for all their children; when
// it cannot take any (pool full of inner nodes) the loop gives up and the caller answers its queries the old way.
// Per-owner operands and results live in `own` (the query's layout); 8-byte LDS atomics there need an 8-byte aligned base.
#pragma once

#include "wost_device.h"

namespace wost {

struct WavePool {
    uint32_t *node, *slot;      // [cap] tasks
    uint32_t *own;              // per-owner words of the query at hand
    int cap;
};
constexpr uint32_t kPoolIndex = (1u << 26) - 1u;      // node and slot indices of a task: trees of up to 11 levels

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS instructions of one wave execute in order: only the compiler has to keep the order
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void pool_push(uint32_t *pool, int &n, bool valid, uint32_t value, int lane)
{
    const unsigned long long mask = __ballot(valid);
    if (valid) pool[n + __popcll(mask & ((1ull << lane) - 1ull))] = value;
    n += __popcll(mask);
}

// node_task(g, owner, leaf, pos, v, key): g = heap index of the node, pos = its index within its level; on a last-level node
// (leaf) it sets v[j] for the slots 4 pos + j to test, else key[j] = (distance bits & ~3) | j for the children to visit
// (0xffffffff = pruned).  slot_task(slot, owner).  All 64 lanes must call; false = gave up (nothing of the answer is valid).
template <class NodeTask, class SlotTask>
__device__ __forceinline__ bool pool_run(const WavePool &W, int levels, bool active, int slot_trigger, NodeTask node_task, SlotTask slot_task)
{
    const int lane = threadIdx.x & 63;
    int n_node = 0, n_slot = 0;
    pool_push(W.node, n_node, active, (uint32_t)lane << 26, lane);        // the roots
    wave_lds_fence();
    while (n_node > 0 || n_slot > 0) {
        if (n_slot >= slot_trigger || n_node == 0) {
            const int k = min(64, n_slot);
            n_slot -= k;
            if (lane < k) {
                const uint32_t e = W.slot[n_slot + lane];
                slot_task(e & kPoolIndex, (int)(e >> 26));
            }
        } else {
            const int k = min(min(64, n_node), (W.cap - n_node) / 3);
            if (k <= 0 || n_slot + 4 * k > W.cap) return false;
            n_node -= k;
            const bool t = lane < k;
            const uint32_t e = t ? W.node[n_node + lane] : 0u;
            wave_lds_fence();        // the tasks are read before the pushes below overwrite them
            const uint32_t own_bits = e & ~kPoolIndex;
            bool leaf = false;
            bool v[4] = {false, false, false, false};
            uint32_t key[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            uint32_t child0 = 0;     // the first child: node index (inner) or leaf slot
            if (t) {
                const uint32_t g = e & kPoolIndex;
                const int level = (31 - __clz((int)(3u * g + 1u))) >> 1;
                const uint32_t pos = g - level_first(level);
                leaf = level == levels;
                child0 = leaf ? 4u * pos : level_first(level + 1) + 4u * pos;
                node_task(g, (int)(e >> 26), leaf, v, key);
                if (!leaf) {
                    cswap(key[0], key[1]); cswap(key[2], key[3]); cswap(key[0], key[2]); cswap(key[1], key[3]); cswap(key[1], key[2]);
                }
            }
            // a last-level node's surviving slots become slot tasks ...
            pool_push(W.slot, n_slot, leaf && v[0], own_bits | (child0 + 0u), lane);
            pool_push(W.slot, n_slot, leaf && v[1], own_bits | (child0 + 1u), lane);
            pool_push(W.slot, n_slot, leaf && v[2], own_bits | (child0 + 2u), lane);
            pool_push(W.slot, n_slot, leaf && v[3], own_bits | (child0 + 3u), lane);
            // ... an inner node's children node tasks, the farthest first: every task's nearest child ends up in the top 64
            pool_push(W.node, n_node, key[3] != 0xffffffffu, own_bits | (child0 + (key[3] & 3u)), lane);
            pool_push(W.node, n_node, key[2] != 0xffffffffu, own_bits | (child0 + (key[2] & 3u)), lane);
            pool_push(W.node, n_node, key[1] != 0xffffffffu, own_bits | (child0 + (key[1] & 3u)), lane);
            pool_push(W.node, n_node, key[0] != 0xffffffffu, own_bits | (child0 + (key[0] & 3u)), lane);
        }
        wave_lds_fence();
    }
    return true;
}

}  // namespace wost
