// Uniform-grid neighbour search core, shared by the KNN (K1 fast path) and radius (N2) kernels.
// Everything here is `__host__ __device__` so the exact same code is exercised on the CPU by
// tests/host_emul (no GPU in the build container) and on gfx950 by knn_grid.hip / radius kernels.
//
// Grid: per cloud, axis-aligned cells of edge cs over the cloud's bounding box, cell id = x + nx*(y + ny*z)
// (+ the cloud's cell_base), supports counting-sorted by cell into `sorted` as float4 {x, y, z, bits(orig idx)}.
// A point's cell coordinate along an axis is floor(u), u = fl(fl(p - origin) * inv_cs) clamped to [0, n-1]; the
// search bounds below are derived in u-space from the SAME float function, so they hold despite rounding:
// a support in cell j has u >= j, and u < j+1 unless j is the last cell.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CBL_HD __host__ __device__ __forceinline__
#else
#define CBL_HD inline
struct float4 { float x, y, z, w; };
#endif

struct CblGrid {            // one per cloud, 48 bytes
    float ox, oy, oz;       // origin = bbox min
    float inv_cs;           // 1 / cell edge
    int nx, ny, nz;         // cells per axis (>= 1)
    int cell_base;          // first global cell id of this cloud
    int start, end;         // support rows [start, end)
    int pad0, pad1;
};

CBL_HD float cbl_u(float p, float o, float inv_cs) { return (p - o) * inv_cs; }
CBL_HD int cbl_cell_coord(float u, int n)
{
    int c = (int)floorf(u);
    c = c < 0 ? 0 : c;
    return c > n - 1 ? n - 1 : c;
}
CBL_HD int cbl_imax(int a, int b) { return a > b ? a : b; }
CBL_HD int cbl_imin(int a, int b) { return a < b ? a : b; }
CBL_HD int cbl_as_int(float f) { union { float f; int i; } v; v.f = f; return v.i; }
CBL_HD float cbl_as_float(int i) { union { float f; int i; } v; v.i = i; return v.f; }

// Choose the grid of a cloud from its bbox [lo,hi], point count and neighbour count k.
// target: ~0.33*k points per cell (the caller's factor, knn_grid.hip) if the cloud filled its bbox uniformly (then the k-th neighbour is
// closer than one cell edge and the 27-cell block certifies most queries); `cap` bounds nx*ny*nz.
CBL_HD void cbl_grid_choose(CblGrid& g, const float lo[3], const float hi[3], int count, float pts_per_cell, int cap)
{
    float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    float emax = fmaxf(ex, fmaxf(ey, ez));
    if (!(emax > 0.f)) emax = 1.f;                       // all points coincide (or empty cloud)
    const float tiny = emax * 1e-3f;                     // flat clouds: give thin axes a nominal thickness
    const float vx = fmaxf(ex, tiny), vy = fmaxf(ey, tiny), vz = fmaxf(ez, tiny);
    float cs = cbrtf(vx * vy * vz * pts_per_cell / (float)(count > 0 ? count : 1));
    cs = fmaxf(cs, emax * (1.0f / 1024.0f));             // <= 1025 cells per axis keeps u-space rounding < 2e-3
    int nx, ny, nz;
    for (int it = 0; it < 64; it++) {
        nx = (int)floorf(ex / cs) + 1; ny = (int)floorf(ey / cs) + 1; nz = (int)floorf(ez / cs) + 1;
        if ((long long)nx * ny * nz <= (long long)cap) break;
        cs *= 1.1f;
    }
    g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
    g.inv_cs = 1.0f / cs;
    g.nx = nx; g.ny = ny; g.nz = nz;
}

// Grid for a RADIUS search: cell edge = 1.001 * radius (enlarged only if the cell budget `cap` is exceeded), so every support
// within `radius` of a query lies in the 27-cell block around the query's cell: a true axis offset < radius is < 0.999 cells,
// and the two u values carry < 2.5e-4 cells of rounding (|u| <= ~1030), so the floor()ed cell coordinates differ by at most 1.
CBL_HD void cbl_grid_choose_radius(CblGrid& g, const float lo[3], const float hi[3], float radius, int cap)
{
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    float emax = fmaxf(ex, fmaxf(ey, ez));
    if (!(emax > 0.f)) emax = 1.f;
    float cs = fmaxf(radius * 1.001f, emax * (1.0f / 1024.0f));
    int nx, ny, nz;
    for (int it = 0; it < 200; it++) {
        nx = (int)floorf(ex / cs) + 1; ny = (int)floorf(ey / cs) + 1; nz = (int)floorf(ez / cs) + 1;
        if ((long long)nx * ny * nz <= (long long)cap) break;
        cs *= 1.1f;
    }
    g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
    g.inv_cs = 1.0f / cs;
    g.nx = nx; g.ny = ny; g.nz = nz;
}

CBL_HD int cbl_cell_of(const CblGrid& g, float x, float y, float z)
{
    const int cx = cbl_cell_coord(cbl_u(x, g.ox, g.inv_cs), g.nx);
    const int cy = cbl_cell_coord(cbl_u(y, g.oy, g.inv_cs), g.ny);
    const int cz = cbl_cell_coord(cbl_u(z, g.oz, g.inv_cs), g.nz);
    return g.cell_base + cx + g.nx * (cy + g.ny * cz);
}

// Lower bound (squared, with safety margins) on the float d2 of every support OUTSIDE the cube of cells
// [c-r, c+r]^3 around the query's cell; +inf if that cube already covers the whole grid.
//   high side of axis a exists if c_a + r + 1 <= n_a - 1 : every such support has u_p >= c_a + r + 1
//   low side exists if c_a - r - 1 >= 0                  : every such support has u_p <  c_a - r
// 0.004 absorbs the rounding of both u values (|u| <= ~1030 cells -> < 2.5e-4 each, taken 8x), 1e-5 the rounding
// of d2 itself and of the division by inv_cs.
CBL_HD float cbl_outside_bound2(const CblGrid& g, float uqx, float uqy, float uqz, int cx, int cy, int cz, int r)
{
    float gap = INFINITY;
    if (cx + r + 1 <= g.nx - 1) gap = fminf(gap, (float)(cx + r + 1) - uqx);
    if (cx - r - 1 >= 0)        gap = fminf(gap, uqx - (float)(cx - r));
    if (cy + r + 1 <= g.ny - 1) gap = fminf(gap, (float)(cy + r + 1) - uqy);
    if (cy - r - 1 >= 0)        gap = fminf(gap, uqy - (float)(cy - r));
    if (cz + r + 1 <= g.nz - 1) gap = fminf(gap, (float)(cz + r + 1) - uqz);
    if (cz - r - 1 >= 0)        gap = fminf(gap, uqz - (float)(cz - r));
    if (gap == INFINITY) return INFINITY;
    gap -= 0.004f;
    if (gap <= 0.f) return 0.f;
    const float d = gap / g.inv_cs;
    return d * d * (1.0f - 1e-5f);
}

// ------------------------------------------------------------------------------------------------------
// Register-resident ascending top-K list (K compile-time: every index is static, nothing goes to scratch).
// Insertion is by strict '<' on d2: among equal distances the order is arbitrary — queries with any tie that
// could matter are detected by `certify` and replayed by the exact kernel, so it never shows.
// ------------------------------------------------------------------------------------------------------
template <int K>
struct CblTopK {
    float d[K];
    int id[K];
    float min_rejected;      // smallest d2 among candidates that are NOT in the list (evicted or refused)
    int count;               // candidates seen (saturating use: only compared with K)

    CBL_HD void init()
    {
#pragma unroll
        for (int j = 0; j < K; j++) { d[j] = INFINITY; id[j] = -1; }
        min_rejected = INFINITY; count = 0;
    }
    CBL_HD float worst() const { return d[K - 1]; }
    CBL_HD void offer(float d2, int i)
    {
        count++;
        if (d2 < d[K - 1]) {
            min_rejected = fminf(min_rejected, d[K - 1]);       // evicted (INFINITY while not full: harmless)
#pragma unroll
            for (int j = K - 1; j >= 0; j--) {
                const bool shift = (j > 0) && (d2 < d[j - 1]);   // element j-1 moves up to j
                if (shift) { d[j] = d[j - 1]; id[j] = id[j - 1]; }
                else if (d2 < d[j]) { d[j] = d2; id[j] = i; }   // first slot whose left neighbour is <= d2
            }
        } else {
            min_rejected = fminf(min_rejected, d2);
        }
    }
    // true if the reference's heap-order-dependent result is fully determined by the set: the list is full,
    // all K distances are distinct and nothing outside the list ties with the K-th.
    CBL_HD bool certify() const
    {
        bool ok = (count >= K) && (min_rejected != d[K - 1]);
#pragma unroll
        for (int j = 1; j < K; j++) ok = ok && (d[j] != d[j - 1]);
        return ok;
    }
};

// One KNN query against one cloud's grid.  Returns true if certified (result final), false if the query
// must be replayed by the exact kernel.  Output rows are written either way.
template <int K>
CBL_HD bool cbl_knn_grid_query(const CblGrid& g, const int* __restrict__ cell_start, const float4* __restrict__ sorted,
                               float qx, float qy, float qz, int k_out, int* __restrict__ idx_row, float* __restrict__ d2_row)
{
    CblTopK<K> top;
    top.init();
    const float uqx = cbl_u(qx, g.ox, g.inv_cs), uqy = cbl_u(qy, g.oy, g.inv_cs), uqz = cbl_u(qz, g.oz, g.inv_cs);
    const int cx = cbl_cell_coord(uqx, g.nx), cy = cbl_cell_coord(uqy, g.ny), cz = cbl_cell_coord(uqz, g.nz);
    auto consider = [&](const float4& s) {
        // same expression as the reference (new - x), knnquery_cuda_kernel.cu:99
        const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        top.offer(d2, cbl_as_int(s.w));
    };
    const int rmax = cbl_imax(cbl_imax(g.nx, g.ny), g.nz);
    bool done = false;
    for (int r = 0; r <= rmax && !done; r++) {
        // shell r: cells with max(|dx|,|dy|,|dz|) == r   (r = 0: the query's own cell).  Rows on the shell's
        // faces take the whole x-range [cx-r, cx+r] as ONE contiguous range of `sorted`; interior rows only the
        // two end cells.  A single inner loop (one inlined copy of the top-K insertion) serves all cases.
        for (int dz = -r; dz <= r; dz++) {
            const int z = cz + dz;
            if (z < 0 || z >= g.nz) continue;
            for (int dy = -r; dy <= r; dy++) {
                const int y = cy + dy;
                if (y < 0 || y >= g.ny) continue;
                const bool full_row = (dz == -r || dz == r || dy == -r || dy == r);
                const int row = g.cell_base + g.nx * (y + g.ny * z);
                const int nseg = full_row ? 1 : 2;
                for (int sg = 0; sg < nseg; sg++) {
                    int x0, x1;
                    if (full_row) { x0 = cbl_imax(cx - r, 0); x1 = cbl_imin(cx + r, g.nx - 1); }
                    else          { x0 = x1 = (sg == 0) ? cx - r : cx + r; }
                    if (x0 < 0 || x1 > g.nx - 1) continue;
                    const int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
                    for (int p = s; p < e; p++) consider(sorted[p]);
                }
            }
        }
        const float bound2 = cbl_outside_bound2(g, uqx, uqy, uqz, cx, cy, cz, r);
        // stop when nothing outside the visited cube can enter the list or tie with its last entry
        done = (bound2 == INFINITY) || (top.count >= K && top.worst() < bound2);
    }
    const bool ok = top.certify();
#pragma unroll
    for (int j = 0; j < K; j++)
        if (j < k_out) { idx_row[j] = top.id[j]; d2_row[j] = top.d[j]; }
    return ok;
}
