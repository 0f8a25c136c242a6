// queryandgroup, ordered form at the networks' full-resolution shapes (C = 32 / 64, K = 8 / 16): the kernel of its own header so that the SAME source is
// also compiled for the host and run with wave semantics by tests/test_gather_pipe_host.py (tests/host_emul/wave provides <hip/hip_runtime.h> and
// "gather_wave.h" for that build).  Included by pointops_gather.hip inside its anonymous namespace.
//   QueryAndGroup   /root/reference/pytorch/lib/pointops/functions/pointops.py:79-100 (grouping of features + centred coordinates, idx given)
#pragma once
#include <gather_wave.h>                                          // angle brackets: the host build of the tests puts its stand-in first on the include path

constexpr int QGP_WAVES = 4;

// The ordered form as persistent, software-pipelined waves.  query_group_lds above starts a fresh wave per piece: sequence slot -> point -> its
// neighbour ids -> their rows -> LDS -> stores, three dependent round trips in front of every piece's stores.  Here a wave walks several
// pieces: the point of the piece after next and the ids of the next piece are fetched a trip ahead, and the NEXT piece's rows are requested
// (into the registers the current piece has just left for LDS) before the current piece's stores are issued.  gfx9 counts loads and stores
// in one counter: requested in that order, the wait for the next rows never waits for the stores behind them, and a wave
// streams stores without ever draining them.  Same bytes, same addresses (bit-identical output).
template <int C4T, int PR>                                         // PR = nsample: rows per piece (every trip count below is a constant: the compiler can count
__global__ __launch_bounds__(256) void query_group_lds_pipe(unsigned npieces, const float* __restrict__ xyz, const float* __restrict__ new_xyz,      // the loads and stores in flight)
                                                            const float4* __restrict__ feat, const int* __restrict__ idx, const int* __restrict__ order,
                                                            float* __restrict__ out)
{
    constexpr int oc = 4 * C4T + 3, NF4 = PR * C4T, NL = (NF4 + 63) / 64, NCH = PR * oc / 4, NS = (NCH + 63) / 64, NC = (3 * PR + 63) / 64;
    static_assert((PR * oc) % 4 == 0 && NF4 % 64 == 0, "pieces are whole 16-byte chunks, feature parts whole wave loads");
    __shared__ __attribute__((aligned(16))) float qg_lds[QGP_WAVES * PR * oc];
    const int lane = threadIdx.x & 63, wv = gw_uniform((int)(threadIdx.x >> 6));
    float* piece = qg_lds + wv * PR * oc;
    const unsigned nwg = (npieces + 3) >> 2, vend = 8 * cbl_xcd_per(nwg), vstep = gridDim.x;
    auto piece_of = [&](unsigned v) -> int {
        const unsigned t = (v < vend ? cbl_xcd_slot(v, nwg) : 0u) * 4 + wv;
        const bool ok = v < vend && t < npieces;
        const int pt = order[ok ? t : 0u];
        return gw_uniform(ok ? pt : -1);
    };
    auto ids_of = [&](int pc) -> int { return idx[(size_t)(pc < 0 ? 0 : pc) * PR + (lane < PR ? lane : 0)]; };
    float4 v[NL]; float cs[NC], cq[NC];
    // rows and coordinates of piece pc (clamped to piece 0 past the end: loaded, never stored); nothing here waits for a load
    auto load_piece = [&](int ids, int pc) {
        const int pq = pc < 0 ? 0 : pc;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int f = lane + 64 * j, row = f / C4T, part = f % C4T;
            v[j] = feat[(size_t)gw_shfl(ids, row) * C4T + part];
        }
#pragma unroll
        for (int h = 0; h < NC; h++) {                             // e = 3 * row + axis (entries past 3 PR: row 0 again, not stored)
            const int e = 64 * h + lane, ec = e < 3 * PR ? e : 0, row = ec / 3, a = ec - 3 * row;
            cs[h] = xyz[(size_t)gw_shfl(ids, row) * 3 + a];
            cq[h] = new_xyz[(size_t)pq * 3 + a];
        }
    };
    // this wave's registers into its LDS piece
    auto park = [&]() {
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int f = lane + 64 * j, row = f / C4T, part = f % C4T;
            float* d = piece + row * oc + 3 + 4 * part;
            d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
        }
#pragma unroll
        for (int h = 0; h < NC; h++) {
            const int e = 64 * h + lane, row = e / 3, a = e - 3 * row;
            if (e < 3 * PR) piece[row * oc + a] = cs[h] - cq[h];
        }
    };
    int p0 = piece_of(blockIdx.x), p1 = piece_of(blockIdx.x + vstep), p2 = piece_of(blockIdx.x + 2 * vstep);
    int id1 = ids_of(p1);
    load_piece(ids_of(p0), p0);
    // Loop invariant at the top of a trip: the LDS piece holds piece `pc`, the registers receive the piece after it.  The trip stores the LDS
    // piece FIRST and parks the registers behind the stores, so the wait for the rows sits in straight-line code behind a known number of
    // stores (a wait at the top of the loop would be merged with the path from the prologue, where no store is in flight, and drain them).
    park();
    int pc = p0;
    p0 = p1; { const int idn = id1; p1 = p2; id1 = ids_of(p1); p2 = piece_of(blockIdx.x + 3 * vstep); load_piece(idn, p0); }
    for (unsigned vv = blockIdx.x; vv < vend; vv += vstep) {
        gw_wave_sync();
        {
            // Unconditional: a trip past the end holds piece 0 (rows, coordinates, centre: all clamped to it) and stores piece 0's own bytes
            // once more (a branch around the stores would again leave a path without them).
            float* o = out + (size_t)(pc < 0 ? 0 : pc) * PR * oc;
#pragma unroll
            for (int u = 0; u < NS; u++) {
                const int ch = lane + 64 * u;
                if (ch < NCH) gw_store16_streaming(o + 4 * ch, piece + 4 * ch);        // 16 aligned bytes from LDS, nontemporal
            }
        }
        gw_wave_sync();                            // LDS is in order: the piece may be rewritten behind the reads above
        park();                                                    // waits for the next piece's rows, not for the stores
        pc = p0;
        p0 = p1; const int idn = id1; p1 = p2;
        id1 = ids_of(p1);
        p2 = piece_of(vv + 4 * vstep);
        load_piece(idn, p0);                                       // the piece after next travels during the next trip's stores
    }
}
