// cell_grid_build.cpp -- host builder of the per-cell candidate lists (see cell_grid.h).
//
// Top-down over a quadtree of square regions: a region inherits the candidate list of its parent,
// measures the exact distance d of its centre to the boundary among those candidates, and keeps
// the chunks whose box comes within d + diagonal of the centre.  Why that is enough: for any point q
// of the region |q - centre| <= R (half the diagonal), so the distance of q is at most d + R, and
// a segment at that distance from q is within d + 2R of the centre -- as is every segment that
// ties with it.  All of this in double precision on the fp32 boxes the device uses, with a
// relative slack far above fp32 rounding: a list can only be too long, never too short.
#include "cell_grid.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace wost {

namespace {

struct Box {
    double cx, cy, ux, uy, hl, hw;
    bool empty;
};

struct Builder {
    const HostTree &t;
    HostCellGrid &g;
    std::vector<Box> boxes;           // per chunk
    double slack_abs;
    // Lists in emission order: the subtrees below quadtree depth kTaskDepth are independent tasks run
    // by a few host threads; a task appends to its own array, a cell remembers task, offset, length.
    struct Task { int x0, y0, size; std::vector<uint16_t> cand; std::vector<uint16_t> ids; };
    std::vector<Task> tasks;
    std::vector<uint32_t> tmp_off, tmp_len;
    std::vector<uint16_t> tmp_task;

    Builder(const HostTree &tree, HostCellGrid &grid) : t(tree), g(grid), slack_abs(0) {}

    static double box_lower(const Box &b, double qx, double qy)
    {
        const double wx = qx - b.cx, wy = qy - b.cy;
        const double u = wx * b.ux + wy * b.uy, v = b.ux * wy - b.uy * wx;
        const double du = std::max(std::fabs(u) - b.hl, 0.0), dv = std::max(std::fabs(v) - b.hw, 0.0);
        return std::sqrt(du * du + dv * dv);
    }

    // exact distance of q to the nearest segment of chunk c
    double chunk_distance(int c, double qx, double qy) const
    {
        double best = std::numeric_limits<double>::infinity();
        const float *r = &g.chunk_seg[(size_t)c * 80];
        for (int j = 0; j < 16; ++j) {
            if (r[j] >= kFarCoord * 0.5f) continue;
            const double wx = qx - r[j], wy = qy - r[16 + j];
            const double ux = r[32 + j], uy = r[48 + j];
            const double u = wx * ux + wy * uy, v = ux * wy - uy * wx;
            const double du = std::max(std::fabs(u) - (double)r[64 + j], 0.0);
            best = std::min(best, std::sqrt(du * du + v * v));
        }
        return best;
    }

    void emit(int task, int ix, int iy, const std::vector<std::pair<double, uint16_t>> &list)
    {
        const uint32_t cell = (uint32_t)iy * (uint32_t)g.nx + (uint32_t)ix;
        std::vector<uint16_t> &ids = tasks[task].ids;
        tmp_task[cell] = (uint16_t)task;
        tmp_off[cell] = (uint32_t)ids.size();
        tmp_len[cell] = (uint32_t)list.size();
        for (const auto &e : list) ids.push_back(e.second);
    }

    // region = cells [x0, x0+size) x [y0, y0+size); cand = candidate chunks of the parent.
    // task < 0: the serial top of the quadtree, which turns regions of `task_size` cells into tasks.
    void descend(int task, int task_size, int x0, int y0, int size, const std::vector<uint16_t> &cand)
    {
        if (x0 >= g.nx || y0 >= g.ny) return;
        const double cxw = (double)g.ox + ((double)x0 + 0.5 * size) * (double)g.h;
        const double cyw = (double)g.oy + ((double)y0 + 0.5 * size) * (double)g.h;
        const double R = 0.5 * std::sqrt(2.0) * size * (double)g.h;
        std::vector<std::pair<double, uint16_t>> lb(cand.size());
        double lb_min = std::numeric_limits<double>::infinity();
        size_t arg = 0;
        for (size_t i = 0; i < cand.size(); ++i) {
            lb[i] = {box_lower(boxes[cand[i]], cxw, cyw), cand[i]};
            if (lb[i].first < lb_min) { lb_min = lb[i].first; arg = i; }
        }
        // exact distance of the centre: nearest box first, then every box that can still beat it
        double d = cand.empty() ? std::numeric_limits<double>::infinity() : chunk_distance(lb[arg].second, cxw, cyw);
        for (size_t i = 0; i < lb.size(); ++i)
            if (i != arg && lb[i].first <= d) d = std::min(d, chunk_distance(lb[i].second, cxw, cyw));
        const double limit = (d + 2.0 * R) * (1.0 + 1e-4) + slack_abs;
        std::vector<std::pair<double, uint16_t>> keep;
        keep.reserve(lb.size());
        for (const auto &e : lb)
            if (e.first <= limit) keep.push_back(e);
        std::vector<uint16_t> next(keep.size());
        for (size_t i = 0; i < keep.size(); ++i) next[i] = keep[i].second;
        if (task < 0 && size <= task_size) {
            tasks.push_back(Task{x0, y0, size, std::move(next), {}});
            return;
        }
        if (size == 1) {
            std::sort(keep.begin(), keep.end());
            emit(task, x0, y0, keep);
            return;
        }
        const int hs = size / 2;
        descend(task, task_size, x0, y0, hs, next);
        descend(task, task_size, x0 + hs, y0, hs, next);
        descend(task, task_size, x0, y0 + hs, hs, next);
        descend(task, task_size, x0 + hs, y0 + hs, hs, next);
    }

    void run_tasks()
    {
        std::atomic<size_t> next_task{0};
        auto worker = [&]() {
            for (;;) {
                const size_t i = next_task.fetch_add(1);
                if (i >= tasks.size()) return;
                Task &tk = tasks[i];
                tk.ids.reserve((size_t)tk.size * tk.size * 10);
                if (tk.size == 1) {
                    // a task of one cell: its candidates are final
                    std::vector<std::pair<double, uint16_t>> keep(tk.cand.size());
                    const double cxw = (double)g.ox + ((double)tk.x0 + 0.5) * (double)g.h, cyw = (double)g.oy + ((double)tk.y0 + 0.5) * (double)g.h;
                    for (size_t k = 0; k < keep.size(); ++k) keep[k] = {box_lower(boxes[tk.cand[k]], cxw, cyw), tk.cand[k]};
                    std::sort(keep.begin(), keep.end());
                    emit((int)i, tk.x0, tk.y0, keep);
                    continue;
                }
                const int hs = tk.size / 2;
                descend((int)i, 0, tk.x0, tk.y0, hs, tk.cand);
                descend((int)i, 0, tk.x0 + hs, tk.y0, hs, tk.cand);
                descend((int)i, 0, tk.x0, tk.y0 + hs, hs, tk.cand);
                descend((int)i, 0, tk.x0 + hs, tk.y0 + hs, hs, tk.cand);
            }
        };
        unsigned n_threads = std::thread::hardware_concurrency();
        n_threads = std::max(1u, std::min(n_threads, 16u));
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < n_threads; ++k) pool.emplace_back(worker);
        worker();
        for (std::thread &th : pool) th.join();
    }
};

}  // namespace

bool build_cell_grid(const HostTree &t, const float lo[2], const float hi[2], int max_cells, HostCellGrid *grid)
{
    *grid = HostCellGrid();
    if (t.n_segs <= 0 || max_cells < 16) return false;
    const int64_t n_chunks = ((int64_t)t.n_segs + 15) / 16;
    if (n_chunks > 65535) return false;
    if (!(hi[0] > lo[0]) || !(hi[1] > lo[1]) || !std::isfinite(lo[0]) || !std::isfinite(hi[0]) || !std::isfinite(lo[1]) ||
        !std::isfinite(hi[1]))
        return false;
    HostCellGrid &g = *grid;
    g.n_chunks = (int32_t)n_chunks;
    // ---- chunks: 16 consecutive occupied slots of the tree's leaf order -------------------------
    Builder B(t, g);
    B.boxes.resize((size_t)n_chunks);
    g.chunk_box.assign((size_t)(n_chunks + 1) * 8, 0.0f);
    {
        float *o = &g.chunk_box[(size_t)n_chunks * 8];      // the sentinel the lists are padded with
        o[0] = kFarCoord; o[1] = kFarCoord; o[2] = 1.0f;
    }
    g.chunk_seg.assign((size_t)(n_chunks + 1) * 80, 0.0f);       // + the sentinel chunk
    g.chunk_slot.assign((size_t)(n_chunks + 1) * 16, -1);
    for (size_t c = 0; c <= (size_t)n_chunks; ++c)
        for (int j = 0; j < 16; ++j) {
            g.chunk_seg[c * 80 + j] = kFarCoord;        // cx
            g.chunk_seg[c * 80 + 16 + j] = kFarCoord;   // cy
            g.chunk_seg[c * 80 + 32 + j] = 1.0f;        // ux
        }
    {
        size_t k = 0;
        std::vector<double> pts;
        const size_t n_slots = t.segOrig.size();
        for (size_t slot = 0; slot <= n_slots; ++slot) {
            const bool occupied = slot < n_slots && t.segOrig[slot] != kFarIndex;
            if (occupied) {
                const size_t c = k / 16, j = k % 16;
                const FlatSeg &s = t.flat[t.segOrig[slot]];
                float *r = &g.chunk_seg[c * 80];
                r[j] = s.cx; r[16 + j] = s.cy; r[32 + j] = s.ux; r[48 + j] = s.uy; r[64 + j] = s.hl;
                g.chunk_slot[c * 16 + j] = (int32_t)slot;
                for (int e = 0; e < 2; ++e) {
                    const SilVertex &v = t.sil[t.segVerts[2 * slot + e]];
                    pts.push_back(v.x); pts.push_back(v.y);
                }
                ++k;
            }
            if (!pts.empty() && ((occupied && k % 16 == 0) || slot == n_slots)) {
                const size_t c = (k - 1) / 16;
                float ob[6];
                fit_obb(pts.data(), pts.size() / 2, t.obb_pad, ob);
                std::memcpy(&g.chunk_box[c * 8], ob, sizeof(ob));
                B.boxes[c] = Box{ob[0], ob[1], ob[2], ob[3], ob[4], ob[5], false};
                pts.clear();
            }
        }
        if ((int64_t)((k + 15) / 16) != n_chunks) return false;
    }
    const int n_used = (int)n_chunks;
    // ---- grid shape ---------------------------------------------------------------------------
    const double ex = (double)hi[0] - lo[0], ey = (double)hi[1] - lo[1];
    double h = std::sqrt(ex * ey / (double)max_cells);
    for (int it = 0; it < 8; ++it) {
        const int64_t nx = (int64_t)std::ceil(ex / h) + 4, ny = (int64_t)std::ceil(ey / h) + 4;
        if (nx * ny <= max_cells) break;
        h *= 1.02;
    }
    g.h = (float)h;
    g.inv_h = 1.0f / g.h;
    g.nx = (int32_t)std::ceil(ex / (double)g.h) + 4;
    g.ny = (int32_t)std::ceil(ey / (double)g.h) + 4;
    g.ox = lo[0] - 2.0f * g.h;
    g.oy = lo[1] - 2.0f * g.h;
    const double ext = std::max(std::max(std::fabs((double)g.ox), std::fabs((double)g.oy)),
                                std::max(std::fabs((double)hi[0]) + 2.0 * h, std::fabs((double)hi[1]) + 2.0 * h));
    // The device finds the cell of q as (int)floor((q - o) * inv_h) in fp32: a point within a few ulp
    // of a cell border may be filed under the neighbouring cell, i.e. lie outside "its" cell by
    // ~ext * 2^-22.  The absolute slack covers that and the fp32 rounding of the distances themselves.
    B.slack_abs = ext * 1e-5 + 1e-30;
    const size_t n_cells = (size_t)g.nx * (size_t)g.ny;
    B.tmp_off.assign(n_cells, 0);
    B.tmp_len.assign(n_cells, 0);
    B.tmp_task.assign(n_cells, 0);
    std::vector<uint16_t> all;
    all.reserve((size_t)n_used);
    for (int c = 0; c < (int)n_chunks; ++c)
        if (!B.boxes[c].empty) all.push_back((uint16_t)c);
    int root = 1;
    while (root < g.nx || root < g.ny) root *= 2;
    // the serial top: regions of root / 16 cells become tasks (at most 256 of them)
    B.descend(-1, std::max(1, root / 16), 0, 0, root, all);
    if (B.tasks.size() > 65535) return false;
    B.run_tasks();
    // ---- row-major lists in groups of four ids + the list of all chunks ------------------------
    g.cell_off.resize(n_cells + 2);
    uint64_t total = 0, groups = 0;
    double mx = 0;
    for (size_t c = 0; c < n_cells; ++c) {
        g.cell_off[c] = (uint32_t)groups;
        total += B.tmp_len[c];
        groups += (B.tmp_len[c] + 3) / 4;
        mx = std::max(mx, (double)B.tmp_len[c]);
    }
    g.cell_off[n_cells] = (uint32_t)groups;
    const uint64_t all_groups = (all.size() + 3) / 4;
    if (groups + all_groups >= (1ull << 32)) return false;
    g.cell_off[n_cells + 1] = (uint32_t)(groups + all_groups);
    g.ids.assign((size_t)(groups + all_groups) * 4, (uint16_t)n_chunks);
    for (size_t c = 0; c < n_cells; ++c)
        if (B.tmp_len[c])
            std::memcpy(&g.ids[(size_t)g.cell_off[c] * 4], B.tasks[B.tmp_task[c]].ids.data() + B.tmp_off[c],
                        (size_t)B.tmp_len[c] * sizeof(uint16_t));
    std::memcpy(&g.ids[(size_t)groups * 4], all.data(), all.size() * sizeof(uint16_t));
    g.mean_list = n_cells ? (double)total / (double)n_cells : 0.0;
    g.max_list = mx;
    g.valid = true;
    return true;
}

}  // namespace wost
