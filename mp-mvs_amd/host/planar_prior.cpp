// planar_prior.cpp -- host construction of the planar prior that feeds the
// second Run() of a Problem:
//   vertices   reference src/PatchMatch.cpp:782-853  GetTriangulateVertices
//   Delaunay   reference src/PatchMatch.cpp:757-780  (cv::Subdiv2D there)
//   plane fit  reference src/PatchMatch.cpp:723-755  (cv::SVD::solveZ there)
//   raster     reference src/PatchMatch.cpp:554-595
// OpenCV is absent from the target image, so the triangulation is an own
// incremental Bowyer-Watson with exact integer predicates, and the 3-point
// null-space solve is the closed form (cross product).  Parity note: for point
// sets with four or more cocircular points (common on a pixel grid) the Delaunay
// triangulation is not unique and cv::Subdiv2D's choice, as well as its triangle
// ORDER (which decides which triangle owns a shared border pixel), cannot be
// reproduced without OpenCV: this file is "parity unpinned" against the
// reference on those two points and pinned by its own property tests.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "PatchMatch.h"

namespace mpmvs_host {

// ---------------------------------------------------------------------------
// vertices: the image is cut into 5x5-pixel cells (smaller at the right / bottom border) and every cell may contribute its
// most reliable pixels (SURVEY a-16).  Two selection rules:
//   plain rule       the pixel of lowest cost, if that cost is a valid one (< 2) and below 0.1;
//   geometric rule   the (up to) three pixels of lowest cost among those with cost < 1 and geometric cost < 0.4, in
//                    ascending cost, as long as their cost stays below max(0.2, 0.85 * cell cost sum / (x_end * y_end)) --
//                    the divisor is the product of the cell's absolute END coordinates, not its area (the reference's
//                    formula, kept because it decides which vertices exist).
// Ties keep the first pixel in raster order.  Both rules are applied by CellPicker below, one cell at a time, cells in
// raster order, which is also the order of the vertex list.
// ---------------------------------------------------------------------------
namespace {
struct CellPicker {
    static constexpr int kCell = 5;
    const float* cost;
    const float* geom;
    int width, height;

    // lowest valid cost of the cell [x0, x1) x [y0, y1) and where it sits
    bool best_pixel(int x0, int y0, int x1, int y1, Point& where, float& lowest) const {
        lowest = 2.0f;
        bool any = false;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const float c = cost[(size_t)y * width + x];
                if (c < 2.0f && c < lowest) {
                    lowest = c;
                    where = Point(x, y);
                    any = true;
                }
            }
        return any;
    }

    // the three lowest costs among the geometrically consistent pixels of the cell, ascending; returns how many pass the
    // cell's adaptive threshold
    int best_three(int x0, int y0, int x1, int y1, Point (&where)[3]) const {
        float low[3] = {2.0f, 2.0f, 2.0f};
        float sum = 0.0f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const size_t i = (size_t)y * width + x;
                const float c = cost[i];
                sum += c;
                if (!(c < 1.0f && geom[i] < 0.4f && c < low[2])) continue;
                // insert into the sorted triple; an equal cost stays behind the earlier pixel
                int slot = 2;
                while (slot > 0 && low[slot - 1] > c) {
                    low[slot] = low[slot - 1];
                    where[slot] = where[slot - 1];
                    --slot;
                }
                low[slot] = c;
                where[slot] = Point(x, y);
            }
        const float scaled = (float)((double)(sum / (float)(y1 * x1)) * 0.85);
        const float limit = scaled > 0.2f ? scaled : 0.2f;
        int n = 0;
        while (n < 3 && low[n] < limit) ++n;
        return n;
    }
};
}  // namespace

void TriangulateVertices(int width, int height, const float* costs, const float* geom_costs, bool geomPlanarPrior,
                         std::vector<Point>& Vertices) {
    Vertices.clear();
    const CellPicker pick{costs, geom_costs, width, height};
    for (int y0 = 0; y0 < height; y0 += CellPicker::kCell)
        for (int x0 = 0; x0 < width; x0 += CellPicker::kCell) {
            const int x1 = std::min(width, x0 + CellPicker::kCell), y1 = std::min(height, y0 + CellPicker::kCell);
            if (geomPlanarPrior) {
                Point p[3];
                const int n = pick.best_three(x0, y0, x1, y1, p);
                Vertices.insert(Vertices.end(), p, p + n);
            } else {
                Point p;
                float lowest;
                if (pick.best_pixel(x0, y0, x1, y1, p, lowest) && lowest < 0.1f) Vertices.push_back(p);
            }
        }
}

// ---------------------------------------------------------------------------
// Delaunay triangulation: incremental Bowyer-Watson, exact 128-bit predicates
// ---------------------------------------------------------------------------
namespace {
typedef __int128 i128;

struct Tri {
    int v[3];  // counter-clockwise
    int n[3];  // n[i]: neighbour across the edge opposite v[i]
};

struct Mesh {
    std::vector<long long> px, py;
    std::vector<Tri> tris;
    std::vector<char> dead;
    int last = 0;
    // scratch reused across inserts (no allocation in the steady state)
    struct Edge {
        int a, b, outer, tri;
    };
    std::vector<int> cavity, stack;
    std::vector<Edge> edges;

    // vertices 0..2 are the far-away super triangle: only predicates touching them
    // need 128 bits, image points (|coord| < 2^15) fit the 64-bit fast path exactly
    i128 orient(int a, int b, int c) const {
        if (a >= 3 && b >= 3 && c >= 3)
            return (i128)((px[b] - px[a]) * (py[c] - py[a]) - (py[b] - py[a]) * (px[c] - px[a]));
        return (i128)(px[b] - px[a]) * (py[c] - py[a]) - (i128)(py[b] - py[a]) * (px[c] - px[a]);
    }
    // > 0: d strictly inside the circumcircle of counter-clockwise (a, b, c)
    bool incircle_pos(int a, int b, int c, int d) const {
        if (a >= 3 && b >= 3 && c >= 3 && d >= 3) {
            const long long ax = px[a] - px[d], ay = py[a] - py[d];
            const long long bx = px[b] - px[d], by = py[b] - py[d];
            const long long cx = px[c] - px[d], cy = py[c] - py[d];
            const long long a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
            return ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx) > 0;
        }
        const i128 ax = px[a] - px[d], ay = py[a] - py[d];
        const i128 bx = px[b] - px[d], by = py[b] - py[d];
        const i128 cx = px[c] - px[d], cy = py[c] - py[d];
        const i128 a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
        return ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx) > 0;
    }
    // sign of orient for image points only (64-bit exact)
    inline long long orient64(int a, int b, int c) const { return (px[b] - px[a]) * (py[c] - py[a]) - (py[b] - py[a]) * (px[c] - px[a]); }
    inline bool right_of(int a, int b, int p) const {
        if (a >= 3 && b >= 3) return orient64(a, b, p) < 0;
        return orient(a, b, p) < 0;
    }
    std::vector<int> free_slots;  // slots of deleted triangles are reused: the live mesh stays ~2N entries (cache resident)
    int locate(int p) {
        int t = last;
        for (size_t guard = 0; guard < tris.size() * 3 + 16; ++guard) {
            bool moved = false;
            for (int i = 0; i < 3; ++i) {
                const int a = tris[t].v[(i + 1) % 3], b = tris[t].v[(i + 2) % 3];
                if (right_of(a, b, p)) {
                    t = tris[t].n[i];
                    moved = true;
                    break;
                }
            }
            if (!moved) return t;
        }
        return t;
    }
    bool insert(int p) {
        const int t0 = locate(p);
        for (int i = 0; i < 3; ++i)
            if (px[tris[t0].v[i]] == px[p] && py[tris[t0].v[i]] == py[p]) return false;  // duplicate point
        // cavity = connected set of triangles whose circumcircle strictly contains p
        cavity.clear();
        stack.clear();
        edges.clear();
        stack.push_back(t0);
        dead[t0] = 2;
        while (!stack.empty()) {
            const int t = stack.back();
            stack.pop_back();
            cavity.push_back(t);
            for (int i = 0; i < 3; ++i) {
                const int nb = tris[t].n[i];
                if (nb < 0 || dead[nb]) continue;
                if (incircle_pos(tris[nb].v[0], tris[nb].v[1], tris[nb].v[2], p)) {
                    dead[nb] = 2;
                    stack.push_back(nb);
                }
            }
        }
        for (int t : cavity)
            for (int i = 0; i < 3; ++i) {
                const int nb = tris[t].n[i];
                if (nb >= 0 && dead[nb] == 2) continue;
                Edge e;
                e.a = tris[t].v[(i + 1) % 3];
                e.b = tris[t].v[(i + 2) % 3];
                e.outer = nb;
                e.tri = -1;
                edges.push_back(e);
                // remember which cavity triangle the outer neighbour pointed to
                if (nb >= 0)
                    for (int j = 0; j < 3; ++j)
                        if (tris[nb].n[j] == t) tris[nb].n[j] = -2 - (int)(edges.size() - 1);
            }
        for (int t : cavity) {
            dead[t] = 1;
            free_slots.push_back(t);
        }
        for (Edge& e : edges) {
            Tri nt;
            nt.v[0] = p;
            nt.v[1] = e.a;
            nt.v[2] = e.b;
            nt.n[0] = e.outer;
            nt.n[1] = nt.n[2] = -1;
            if (!free_slots.empty()) {
                e.tri = free_slots.back();
                free_slots.pop_back();
                tris[e.tri] = nt;
                dead[e.tri] = 0;
            } else {
                e.tri = (int)tris.size();
                tris.push_back(nt);
                dead.push_back(0);
            }
        }
        for (size_t k = 0; k < edges.size(); ++k) {
            const Edge& e = edges[k];
            if (e.outer >= 0)
                for (int j = 0; j < 3; ++j)
                    if (tris[e.outer].n[j] == -2 - (int)k) tris[e.outer].n[j] = e.tri;
            for (const Edge& f : edges) {
                if (f.a == e.b) tris[e.tri].n[1] = f.tri;  // across edge (b, p)
                if (f.b == e.a) tris[e.tri].n[2] = f.tri;  // across edge (p, a)
            }
        }
        last = edges.empty() ? last : edges[0].tri;
        return true;
    }
};
}  // namespace

std::vector<Triangle> Delaunay(const Rect boundRC, const std::vector<Point>& points) {
    std::vector<Triangle> results;
    if (points.empty()) return results;
    Mesh m;
    const long long K = 1LL << 24;  // super triangle far outside any image (exactness: |coord| < 2^26)
    (void)boundRC;
    m.px = {-K, 3 * K, -K};
    m.py = {-K, -K, 3 * K};
    m.px.reserve(points.size() + 3);
    m.py.reserve(points.size() + 3);
    m.tris.reserve(points.size() * 2 + 64);
    m.dead.reserve(points.size() * 2 + 64);
    Tri t0;
    t0.v[0] = 0;
    t0.v[1] = 1;
    t0.v[2] = 2;
    t0.n[0] = t0.n[1] = t0.n[2] = -1;
    m.tris.push_back(t0);
    m.dead.push_back(0);
    // Insertion order: the vertices arrive sorted by 5x5 cell in row-major order,
    // which makes every new point fall just outside the current hull and blows the
    // Bowyer-Watson cavities up (measured: 23 triangles per insert).  A biased
    // multi-level order -- cells on a stride-8 lattice first, then stride 4, 2, 1,
    // row-major inside each level -- keeps cavities at ~5 triangles and walks
    // short.  Deterministic; the triangulation itself does not depend on the order
    // (up to cocircular ties).
    std::vector<int> order(points.size());
    for (size_t i = 0; i < points.size(); ++i) order[i] = (int)i;
    auto level = [&](int i) {
        const int cx = points[i].x / 5, cy = points[i].y / 5;
        if (cx % 8 == 0 && cy % 8 == 0) return 0;
        if (cx % 4 == 0 && cy % 4 == 0) return 1;
        if (cx % 2 == 0 && cy % 2 == 0) return 2;
        return 3;
    };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return level(a) < level(b); });
    for (const Point& p : points) {
        m.px.push_back(p.x);
        m.py.push_back(p.y);
    }
    for (int i : order) m.insert(i + 3);
    for (size_t t = 0; t < m.tris.size(); ++t) {
        if (m.dead[t]) continue;
        const Tri& T = m.tris[t];
        if (T.v[0] < 3 || T.v[1] < 3 || T.v[2] < 3) continue;  // touches the super triangle
        results.push_back(Triangle(Point((int)m.px[T.v[0]], (int)m.py[T.v[0]]), Point((int)m.px[T.v[1]], (int)m.py[T.v[1]]),
                                   Point((int)m.px[T.v[2]], (int)m.py[T.v[2]])));
    }
    return results;
}

// ---------------------------------------------------------------------------
// plane through the three back-projected triangle vertices
// ---------------------------------------------------------------------------
static void point_on_ref_cam(const Camera& cam, int x, int y, float depth, double X[3]) {
    // reference src/PatchMatch.cpp:200-209 (Get3DPointonRefCam), fp32 there
    X[0] = (double)(depth * ((float)x - cam.K[2]) / cam.K[0]);
    X[1] = (double)(depth * ((float)y - cam.K[5]) / cam.K[4]);
    X[2] = (double)depth;
}

float4 PriorPlane(const Camera& cam, const Triangle& t, const float4* planes, int width) {
    double X1[3], X2[3], X3[3];
    point_on_ref_cam(cam, t.pt1.x, t.pt1.y, planes[(size_t)t.pt1.y * width + t.pt1.x].w, X1);
    point_on_ref_cam(cam, t.pt2.x, t.pt2.y, planes[(size_t)t.pt2.y * width + t.pt2.x].w, X2);
    point_on_ref_cam(cam, t.pt3.x, t.pt3.y, planes[(size_t)t.pt3.y * width + t.pt3.x].w, X3);
    const double u[3] = {X2[0] - X1[0], X2[1] - X1[1], X2[2] - X1[2]};
    const double v[3] = {X3[0] - X1[0], X3[1] - X1[1], X3[2] - X1[2]};
    double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    double norm = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(norm > 1e-30)) {  // collinear points: no unique plane; take the fronto-parallel one
        n[0] = 0.0;
        n[1] = 0.0;
        n[2] = -1.0;
        norm = 1.0;
    }
    double d = -(n[0] * X1[0] + n[1] * X1[1] + n[2] * X1[2]);
    if (d < 0) norm = -norm;  // reference src/PatchMatch.cpp:746-752: offset made positive
    float4 o;
    o.x = (float)(n[0] / norm);
    o.y = (float)(n[1] / norm);
    o.z = (float)(n[2] / norm);
    o.w = (float)(d / norm);
    return o;
}

static float depth_from_plane(const Camera& cam, const float4 pl, int x, int y) {
    // reference src/PatchMatch.cpp:650-653
    return -pl.w * cam.K[0] / (((float)x - cam.K[2]) * pl.x + (cam.K[0] / cam.K[4]) * ((float)y - cam.K[5]) * pl.y + cam.K[0] * pl.z);
}

void BuildPrior(const Camera& cam, int width, int height, const std::vector<Triangle>& triangles, const float4* planes,
                float depth_min, float depth_max, std::vector<float4>& planeParams, Image& mask) {
    const Rect imageRC{0, 0, width, height};
    // triangles the reference keeps (all three vertices inside the image), in order:
    // triangle k gets label k + 1 and its plane is planeParams[k]
    std::vector<int> keep;
    keep.reserve(triangles.size());
    for (size_t i = 0; i < triangles.size(); ++i) {
        const Triangle& t = triangles[i];
        if (imageRC.contains(t.pt1) && imageRC.contains(t.pt2) && imageRC.contains(t.pt3)) keep.push_back((int)i);
    }
    planeParams.assign(keep.size(), float4{0, 0, 0, 0});
    // A pixel covered by several triangles keeps the label of the LAST one in the
    // reference's sequential loop = the largest label: an atomic max over triangles
    // processed in parallel gives the same mask.
    std::vector<uint32_t> label((size_t)width * height, 0u);
#pragma omp parallel for schedule(dynamic, 256)
    for (long k = 0; k < (long)keep.size(); ++k) {
        const Triangle& t = triangles[keep[k]];
        const uint32_t lab = (uint32_t)k + 1u;
        const float L01 = (float)std::sqrt((double)((t.pt1.x - t.pt2.x) * (t.pt1.x - t.pt2.x) + (t.pt1.y - t.pt2.y) * (t.pt1.y - t.pt2.y)));
        const float L02 = (float)std::sqrt((double)((t.pt1.x - t.pt3.x) * (t.pt1.x - t.pt3.x) + (t.pt1.y - t.pt3.y) * (t.pt1.y - t.pt3.y)));
        const float L12 = (float)std::sqrt((double)((t.pt2.x - t.pt3.x) * (t.pt2.x - t.pt3.x) + (t.pt2.y - t.pt3.y) * (t.pt2.y - t.pt3.y)));
        const float max_edge = std::max(L01, std::max(L02, L12));
        const float step = (float)(1.0 / max_edge);
        // barycentric stepping of the reference (src/PatchMatch.cpp:564-570)
        for (float p = 0; p < 1.0; p += step) {
            for (float q = 0; q < 1.0 - p; q += step) {
                const int x = (int)((double)(p * (float)t.pt1.x + q * (float)t.pt2.x) + (1.0 - p - q) * t.pt3.x);
                const int y = (int)((double)(p * (float)t.pt1.y + q * (float)t.pt2.y) + (1.0 - p - q) * t.pt3.y);
                if (x >= 0 && y >= 0 && x < width && y < height) {
                    uint32_t* cell = &label[(size_t)y * width + x];
                    uint32_t cur = __atomic_load_n(cell, __ATOMIC_RELAXED);
                    while (cur < lab && !__atomic_compare_exchange_n(cell, &cur, lab, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                    }
                }
            }
            if (!(step > 0.0f) || !std::isfinite(step)) break;  // degenerate triangle: one sample
        }
        planeParams[k] = PriorPlane(cam, t, planes, width);
    }
    mask = Image(height, width, 1, 0.0f);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < height; ++j)
        for (int i = 0; i < width; ++i) {
            const uint32_t lab = label[(size_t)j * width + i];
            if (lab > 0) {
                const float d = depth_from_plane(cam, planeParams[lab - 1], i, j);
                if (d <= depth_max && d >= depth_min) mask.at(j, i) = (float)lab;
            }
        }
}

}  // namespace mpmvs_host
