// nd_symbolic.hpp -- symbolic phase of the nested-dissection (multifrontal) Cholesky of a stencil matrix on an
// M x N pixel grid.  Host only, plain C++17 (no HIP): the same file is compiled into libbpltv (nd_solver.hpp) and
// into the host check tools/nd_host_check.cpp (g++).
//
// What it is for: the reduced adjoint systems behind Julia's `\` at /root/reference/src/TVLearningFunctionVec.jl:131,
// 248 (7-point stencil: offsets 1, M-1, M) and /root/reference/src/SumRegsLearningFunction.jl:324,394 (13-point
// stencil: 1, 2, M-1, M, M+1, 2M) are sparse SPD matrices on the pixel grid.  A banded Cholesky costs n bw^2 =
// O(M^4) flop and n bw entries (8.6 GB for 1024^2); nested dissection costs O(M^3) (George: 9.9 M^3 flop,
// 7.75 M^2 log2 M fill -- ~1e10 flop and < 1 GB for 1024^2).
//
// Structure: the grid is cut recursively by separator lines of `sepw` pixels (the stencil's reach: 1 for the TV
// model, 2 for the sum of regularisers) across its longer side; a rectangle of at most `leaf_pix` pixels is a
// leaf whose pixels are all eliminated at once.  Every tree node is one dense FRONT: its `p` pivots (the
// separator, or the leaf's pixels) followed by its `b` boundary pixels -- the pixels outside the node's rectangle
// that the stencil couples to it, all of which are pivots of ancestors.  Fronts are ordered by elimination index,
// so a child's boundary maps monotonically into its parent's front (cmap) and only lower triangles are needed.
// Nodes are numbered level by level (depth from the root): level l is eliminated after level l + 1, all its fronts
// (of all images) in the same launches.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

namespace bpltv {

struct NdStencil {
    int nd = 0;          // planes (diagonals) of the assembled matrix, plane 0 = main diagonal
    int di[8] = {0}, dj[8] = {0};   // plane t couples pixel (i, j) with (i + di, j + dj); linear offset di + M dj > 0 for t > 0
    int reach = 1;       // max(|di|, |dj|): separator width
};
// TV model: adj_assemble_kernel's band4 = offsets {0, 1, M-1, M}
inline NdStencil nd_stencil_tv() {
    NdStencil s; s.nd = 4; s.reach = 1;
    const int d[4][2] = {{0, 0}, {1, 0}, {-1, 1}, {0, 1}};
    for (int t = 0; t < 4; ++t) { s.di[t] = d[t][0]; s.dj[t] = d[t][1]; }
    return s;
}
// sum of regularisers: sr_adj_assemble_kernel's seven diagonals = offsets {0, 1, 2, M-1, M, M+1, 2M}
inline NdStencil nd_stencil_sr() {
    NdStencil s; s.nd = 7; s.reach = 2;
    const int d[7][2] = {{0, 0}, {1, 0}, {2, 0}, {-1, 1}, {0, 1}, {1, 1}, {0, 2}};
    for (int t = 0; t < 7; ++t) { s.di[t] = d[t][0]; s.dj[t] = d[t][1]; }
    return s;
}

struct NdNode {
    int level = 0, parent = -1;
    int child[2] = {-1, -1};
    int i0 = 0, i1 = 0, j0 = 0, j1 = 0;   // rectangle of the subtree (pivots of this node and of all descendants)
    int p = 0, b = 0;                     // pivots, boundary pixels; front order = pivots, then boundary
    int piv_off = 0;                      // pix[piv_off .. +p): pivot pixels (linear index i + M j), then
                                          // pix[piv_off + p .. +b): boundary pixels, sorted by elimination index
    int cmap_off = 0;                     // cmap[cmap_off .. +b): boundary entry k -> front-local index in the parent
    int orig_off = 0, orig_cnt = 0;       // matrix entries assembled into this front (NdOrig)
    int64_t fac_off = 0;                  // doubles: factor columns (f x p, leading dimension f = p + b) of one image
    int64_t u_off = 0;                    // doubles: update matrix (b x b, leading dimension b) in its level's workspace
    int64_t uv_off = 0;                   // doubles: update vector (b) of the substitutions, all levels in one array
};
// F(r, c) += planes[plane & 15][pixel], r >= c front-local.  Bit ND_ORIG_UPPER of `plane` says that the entry F(r, c) =
// A[g_r][g_c] has the smaller pixel as its ROW (g_r < g_c: an entry of A's upper triangle, stored at pixel g_r) -- the
// same number for a symmetric matrix; the LU variant (nd_solver.hpp, lu = true) reads it from the upper diagonals, and
// the mirror entry F(c, r) from the other set.
constexpr int ND_ORIG_UPPER = 16;
struct NdOrig { int r, c, plane, pixel; };

struct NdTree {
    int M = 0, N = 0, n = 0;
    NdStencil st;
    std::vector<NdNode> nodes;            // level by level, root = node 0
    std::vector<int> lvl_start;           // nodes of level l: [lvl_start[l], lvl_start[l+1])
    std::vector<int> pix, cmap;
    std::vector<NdOrig> orig;
    std::vector<int> elim;                // elimination index of every pixel
    int64_t fac_doubles = 0;              // factor storage per image
    int64_t ws_doubles[2] = {0, 0};       // update-matrix workspace per image for even / odd levels
    int64_t uv_doubles = 0;               // update vectors per image
    int max_f = 0, max_p = 0;
    bool ok = true;                       // false: an invariant of the construction failed (err says which); the tree is unusable
    std::string err;
    int levels() const { return (int)lvl_start.size() - 1; }
    double flops() const {                // of one factorisation (multiply-adds counted as 2)
        double s = 0;
        for (const NdNode& v : nodes) { const double p = v.p, b = v.b; s += p * p * p / 3 + p * p * b + p * b * b; }
        return s;
    }
};

// Build the tree for an M x N grid.  leaf_pix: rectangles of at most this many pixels become leaves.
inline NdTree nd_build(int M, int N, const NdStencil& st, int leaf_pix = 32) {
    NdTree T;
    T.M = M; T.N = N; T.n = M * N; T.st = st;
    const int w = st.reach;
    struct Tmp { int i0, i1, j0, j1, parent, level, child[2]; int si0, si1, sj0, sj1; bool leaf; };
    std::vector<Tmp> tmp;
    // breadth-first: the node array comes out level by level
    tmp.push_back({0, M, 0, N, -1, 0, {-1, -1}, 0, 0, 0, 0, false});
    for (size_t q = 0; q < tmp.size(); ++q) {
        Tmp t = tmp[q];
        const int h = t.i1 - t.i0, wd = t.j1 - t.j0;
        // a rectangle too thin to leave two non-empty halves beside a separator is a leaf as well
        const bool along_i = h >= wd;               // cut the longer side
        const int len = along_i ? h : wd;
        if ((int64_t)h * wd <= leaf_pix || len < w + 2) {
            t.leaf = true;
            t.si0 = t.i0; t.si1 = t.i1; t.sj0 = t.j0; t.sj1 = t.j1;
            tmp[q] = t;
            continue;
        }
        const int s = (len - w) / 2;                // first half: [0, s), separator [s, s + w), second half [s + w, len)
        Tmp a = t, b = t;
        a.parent = b.parent = (int)q; a.level = b.level = t.level + 1;
        a.child[0] = a.child[1] = b.child[0] = b.child[1] = -1;
        if (along_i) {
            t.si0 = t.i0 + s; t.si1 = t.si0 + w; t.sj0 = t.j0; t.sj1 = t.j1;
            a.i1 = t.si0; b.i0 = t.si1;
        } else {
            t.sj0 = t.j0 + s; t.sj1 = t.sj0 + w; t.si0 = t.i0; t.si1 = t.i1;
            a.j1 = t.sj0; b.j0 = t.sj1;
        }
        t.child[0] = (int)tmp.size(); t.child[1] = (int)tmp.size() + 1;
        tmp[q] = t;
        tmp.push_back(a);
        tmp.push_back(b);
    }
    const int nn = (int)tmp.size();
    T.nodes.resize(nn);
    int maxlvl = 0;
    for (int q = 0; q < nn; ++q) maxlvl = std::max(maxlvl, tmp[q].level);
    T.lvl_start.assign(maxlvl + 2, 0);
    for (int q = 0; q < nn; ++q) T.lvl_start[tmp[q].level + 1]++;
    for (int l = 0; l <= maxlvl; ++l) T.lvl_start[l + 1] += T.lvl_start[l];
    // elimination order: deepest level first, within a level by node number, within a node along the separator
    // (column by column: linear pixel order).  Any order with children before parents gives the same fronts.
    T.elim.assign(T.n, -1);
    int next = 0;
    for (int q = 0; q < nn; ++q) {
        NdNode& v = T.nodes[q];
        const Tmp& t = tmp[q];
        v.level = t.level; v.parent = t.parent; v.child[0] = t.child[0]; v.child[1] = t.child[1];
        v.i0 = t.i0; v.i1 = t.i1; v.j0 = t.j0; v.j1 = t.j1;
        v.p = (t.si1 - t.si0) * (t.sj1 - t.sj0);
    }
    for (int l = maxlvl; l >= 0; --l)
        for (int q = T.lvl_start[l]; q < T.lvl_start[l + 1]; ++q) {
            const Tmp& t = tmp[q];
            for (int j = t.sj0; j < t.sj1; ++j)
                for (int i = t.si0; i < t.si1; ++i) T.elim[i + M * j] = next++;
        }
    // fronts: pivots, then the boundary ring sorted by elimination index
    std::vector<int> stamp(T.n, -1), pos(T.n, -1), ring;
    for (int q = 0; q < nn; ++q) {
        NdNode& v = T.nodes[q];
        const Tmp& t = tmp[q];
        v.piv_off = (int)T.pix.size();
        for (int j = t.sj0; j < t.sj1; ++j)
            for (int i = t.si0; i < t.si1; ++i) T.pix.push_back(i + M * j);
        ring.clear();
        auto visit = [&](int i, int j) {           // neighbours of rectangle pixel (i, j) outside the rectangle
            for (int s = 1; s < st.nd; ++s)
                for (int sg = -1; sg <= 1; sg += 2) {
                    const int a = i + sg * st.di[s], c = j + sg * st.dj[s];
                    if (a < 0 || a >= M || c < 0 || c >= N) continue;
                    if (a >= v.i0 && a < v.i1 && c >= v.j0 && c < v.j1) continue;
                    const int g = a + M * c;
                    if (stamp[g] != q) { stamp[g] = q; ring.push_back(g); }
                }
        };
        for (int j = v.j0; j < v.j1; ++j) {
            const bool edge_j = (j - v.j0 < w) || (v.j1 - 1 - j < w);
            for (int i = v.i0; i < v.i1; ++i) {
                if (!edge_j && (i - v.i0 >= w) && (v.i1 - 1 - i >= w)) { i = std::max(i, v.i1 - w - 1); continue; }
                visit(i, j);
            }
        }
        std::sort(ring.begin(), ring.end(), [&](int a, int c) { return T.elim[a] < T.elim[c]; });
        v.b = (int)ring.size();
        for (int g : ring) T.pix.push_back(g);
        T.max_f = std::max(T.max_f, v.p + v.b);
        T.max_p = std::max(T.max_p, v.p);
    }
    // child boundary -> parent front, matrix entries per front, storage offsets
    T.cmap.reserve(T.pix.size());
    int64_t uv = 0;
    for (int l = 0; l <= maxlvl; ++l) {
        int64_t uoff = 0;
        for (int q = T.lvl_start[l]; q < T.lvl_start[l + 1]; ++q) {
            NdNode& v = T.nodes[q];
            v.fac_off = T.fac_doubles;
            T.fac_doubles += (int64_t)(v.p + v.b) * v.p;
            v.u_off = uoff;
            uoff += (int64_t)v.b * v.b;
            v.uv_off = uv;
            uv += v.b;
        }
        T.ws_doubles[l & 1] = std::max(T.ws_doubles[l & 1], uoff);
    }
    T.uv_doubles = uv;
    for (int q = 0; q < nn; ++q) {
        NdNode& v = T.nodes[q];
        const int f = v.p + v.b;
        for (int k = 0; k < f; ++k) pos[T.pix[v.piv_off + k]] = k;
        // entries A(q1, q2) with q1 a pivot of this front and q2 in the front, not eliminated before q1
        v.orig_off = (int)T.orig.size();
        for (int k = 0; k < v.p; ++k) {
            const int g = T.pix[v.piv_off + k], gi = g % M, gj = g / M;
            T.orig.push_back({k, k, 0, g});
            for (int s = 1; s < st.nd; ++s)
                for (int sg = -1; sg <= 1; sg += 2) {
                    const int a = gi + sg * st.di[s], c = gj + sg * st.dj[s];
                    if (a < 0 || a >= M || c < 0 || c >= N) continue;
                    const int g2 = a + M * c;
                    if (T.elim[g2] < T.elim[g]) continue;           // assembled where g2 was a pivot (or it is k' < k here)
                    const int k2 = pos[g2];                          // in this front by construction
                    if (k2 < 0 || k2 >= f || T.pix[v.piv_off + k2] != g2) {
                        T.ok = false; T.err = "matrix entry of front " + std::to_string(q) + " couples a pixel outside the front";
                        return T;
                    }
                    // plane s stores A[c + off][c] at the pixel of smaller linear index
                    T.orig.push_back({k2, k, sg > 0 ? s : (s | ND_ORIG_UPPER), sg > 0 ? g : g2});
                }
        }
        v.orig_cnt = (int)T.orig.size() - v.orig_off;
        for (int ci = 0; ci < 2; ++ci) {
            if (v.child[ci] < 0) continue;
            NdNode& ch = T.nodes[v.child[ci]];
            ch.cmap_off = (int)T.cmap.size();
            for (int k = 0; k < ch.b; ++k) {
                const int g = T.pix[ch.piv_off + ch.p + k];
                const int k2 = pos[g];
                if (k2 < 0 || k2 >= f || T.pix[v.piv_off + k2] != g) {   // the child's boundary lies in the parent's front
                    T.ok = false; T.err = "boundary of front " + std::to_string(v.child[ci]) + " is not contained in its parent's front";
                    return T;
                }
                T.cmap.push_back(k2);
            }
        }
    }
    return T;
}

}  // namespace bpltv
