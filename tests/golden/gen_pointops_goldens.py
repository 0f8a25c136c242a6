#!/usr/bin/env python3
"""Generate tests/golden/pointops_*.npz from the reference's own pointops kernel bodies.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).  The fixtures it writes are data only
(inputs + outputs); no reference source text is written into the repository.

How: the reference ships pointops as CUDA only (pytorch/lib/pointops/src/*/*_cuda_kernel.cu) and
this image has no CUDA toolkit, so the .cu files cannot be built as they are.  Their kernel *bodies*
are plain C++ once the CUDA execution context exists, so this script builds a throw-away harness in
a temp dir that (1) provides that context on the host — blockIdx/threadIdx/blockDim variables, a
serial atomicAdd, __syncthreads() as a cooperative-fiber yield — (2) #includes a temp copy of each
.cu with its `#include` lines and its `<<<...>>>` launcher (the tail of the file) cut off, and
(3) runs every (block, thread) of the launch the reference's launcher would have made
(blocks = ceil(work/256), 256 threads; FPS: one block per cloud of opt_n_threads(n_max) threads).

Threads of a block run as ucontext fibers, one at a time, each until its next __syncthreads();
within a barrier interval fibers run in DESCENDING thread order so that thread 0 — which overwrites
dists_i[0] early in the next FPS iteration — runs after every other thread has read it
(sampling_cuda_kernel.cu:125 vs :60-61; on the GPU the long strided loop hides that race).

Usage: python tests/golden/gen_pointops_goldens.py
"""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference/pytorch/lib/pointops/src"
OUT = os.path.dirname(os.path.abspath(__file__))

SHIM = r'''
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <ucontext.h>
#define __global__
#define __device__
#define __host__
#define __shared__ static
struct uint3_ { unsigned x, y, z; };
static uint3_ blockIdx, threadIdx, blockDim, gridDim;
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, float b) { return fminf(a, b); }

// ---- cooperative fibers for kernels that use __syncthreads() --------------------------------
static ucontext_t sched_ctx;
static std::vector<ucontext_t> fib_ctx;
static std::vector<char> fib_done;
static int fib_cur = -1;
static inline void __syncthreads() { swapcontext(&fib_ctx[fib_cur], &sched_ctx); }
'''

DRIVER = r'''
#define THREADS_PER_BLOCK 256
#define FOR_EACH_THREAD(total)                                                   \
    blockDim.x = THREADS_PER_BLOCK; gridDim.x = ((total) + 255) / 256;           \
    for (blockIdx.x = 0; blockIdx.x < gridDim.x; blockIdx.x++)                   \
        for (threadIdx.x = 0; threadIdx.x < blockDim.x; threadIdx.x++)

extern "C" {
void ref_knnquery(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset, int* idx, float* dist2)
{ FOR_EACH_THREAD(m) knnquery_cuda_kernel(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2); }
void ref_grouping_forward(int m, int nsample, int c, const float* input, const int* idx, float* output)
{ FOR_EACH_THREAD(m * nsample * c) grouping_forward_cuda_kernel(m, nsample, c, input, idx, output); }
void ref_grouping_backward(int m, int nsample, int c, const float* go, const int* idx, float* gi)
{ FOR_EACH_THREAD(m * nsample * c) grouping_backward_cuda_kernel(m, nsample, c, go, idx, gi); }
void ref_interpolation_forward(int n, int c, int k, const float* input, const int* idx, const float* w, float* out)
{ FOR_EACH_THREAD(n * c) interpolation_forward_cuda_kernel(n, c, k, input, idx, w, out); }
void ref_interpolation_backward(int n, int c, int k, const float* go, const int* idx, const float* w, float* gi)
{ FOR_EACH_THREAD(n * c) interpolation_backward_cuda_kernel(n, c, k, go, idx, w, gi); }
void ref_subtraction_forward(int n, int ns, int c, const float* a, const float* b, const int* idx, float* out)
{ FOR_EACH_THREAD(n * ns * c) subtraction_forward_cuda_kernel(n, ns, c, a, b, idx, out); }
void ref_subtraction_backward(int n, int ns, int c, const int* idx, const float* go, float* g1, float* g2)
{ FOR_EACH_THREAD(n * ns * c) subtraction_backward_cuda_kernel(n, ns, c, idx, go, g1, g2); }
void ref_aggregation_forward(int n, int ns, int c, int wc, const float* in, const float* pos, const float* w, const int* idx, float* out)
{ FOR_EACH_THREAD(n * c) aggregation_forward_cuda_kernel(n, ns, c, wc, in, pos, w, idx, out); }
void ref_aggregation_backward(int n, int ns, int c, int wc, const float* in, const float* pos, const float* w, const int* idx, const float* go, float* gi, float* gp, float* gw)
{ FOR_EACH_THREAD(n * c) aggregation_backward_cuda_kernel(n, ns, c, wc, in, pos, w, idx, go, gi, gp, gw); }
}

// ---- FPS: one block per cloud, B threads as fibers ------------------------------------------
struct FpsArgs { const float* xyz; const int* offset; const int* new_offset; float* tmp; int* idx; };
static FpsArgs fps_args;
template <unsigned B> static void fps_entry()
{
    furthestsampling_cuda_kernel<B>(fps_args.xyz, fps_args.offset, fps_args.new_offset, fps_args.tmp, fps_args.idx);
    fib_done[fib_cur] = 1;
    swapcontext(&fib_ctx[fib_cur], &sched_ctx);
}
template <unsigned B> static void fps_run_block()
{
    const size_t STK = 64 * 1024;
    std::vector<char> stacks(STK * B);
    fib_ctx.assign(B, ucontext_t()); fib_done.assign(B, 0);
    blockDim.x = B;
    for (unsigned t = 0; t < B; t++) {
        getcontext(&fib_ctx[t]);
        fib_ctx[t].uc_stack.ss_sp = &stacks[STK * t]; fib_ctx[t].uc_stack.ss_size = STK; fib_ctx[t].uc_link = &sched_ctx;
        makecontext(&fib_ctx[t], (void (*)())fps_entry<B>, 0);
    }
    for (;;) {
        bool any = false;
        for (int t = (int)B - 1; t >= 0; t--) {           // descending: thread 0 last (see module docstring)
            if (fib_done[t]) continue;
            any = true; fib_cur = t; threadIdx.x = t;
            swapcontext(&sched_ctx, &fib_ctx[t]);
        }
        if (!any) break;
    }
}
extern "C" int ref_opt_n_threads(int n) { return opt_n_threads(n); }
extern "C" void ref_furthestsampling(int b, int n, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx)
{
    fps_args = FpsArgs{xyz, offset, new_offset, tmp, idx};
    const unsigned B = opt_n_threads(n);
    gridDim.x = b;
    for (blockIdx.x = 0; blockIdx.x < (unsigned)b; blockIdx.x++) {
        switch (B) {
            case 1024: fps_run_block<1024>(); break; case 512: fps_run_block<512>(); break;
            case 256: fps_run_block<256>(); break;   case 128: fps_run_block<128>(); break;
            case 64: fps_run_block<64>(); break;     case 32: fps_run_block<32>(); break;
            case 16: fps_run_block<16>(); break;     case 8: fps_run_block<8>(); break;
            case 4: fps_run_block<4>(); break;       case 2: fps_run_block<2>(); break;
            case 1: fps_run_block<1>(); break;       default: fps_run_block<512>(); break;   // launcher default, :168
        }
    }
}
'''


def strip_cu(path):
    """temp copy of a reference .cu: drop #include lines and everything from the first launcher on."""
    src = open(path).read()
    m = re.search(r"^void \w+_launcher", src, flags=re.M)
    body = src[: m.start()] if m else src
    return "\n".join(l for l in body.splitlines() if not l.lstrip().startswith("#include"))


def build_ref(tmp):
    parts = [SHIM]
    # cuda_utils.h defines opt_n_threads (needs no CUDA types except dim3 in one unused helper)
    cu = open(os.path.join(REF, "cuda_utils.h")).read()
    cu = cu.replace("#define THREADS_PER_BLOCK 256", "")
    cu = re.sub(r"inline dim3 opt_block_config.*?\n}\n", "", cu, flags=re.S)
    open(os.path.join(tmp, "cuda_utils_tmp.h"), "w").write(cu)
    parts.append('#include "cuda_utils_tmp.h"\n')
    for op in ["knnquery", "sampling", "grouping", "interpolation", "subtraction", "aggregation"]:
        name = f"{op}_body_tmp.inc"
        open(os.path.join(tmp, name), "w").write(strip_cu(os.path.join(REF, op, f"{op}_cuda_kernel.cu")))
        parts.append(f'#include "{name}"\n')
    parts.append(DRIVER)
    open(os.path.join(tmp, "harness.cpp"), "w").write("".join(parts))
    so = os.path.join(tmp, "libref_pointops_host.so")
    # -O1: keep fibers simple; -ffp-contract=off: host semantics (see DESIGN.md "FMA contraction")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-fPIC", "-ffp-contract=off", "-shared", "-o", so,
                           os.path.join(tmp, "harness.cpp")])
    return ctypes.CDLL(so)


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def knn(lib, xyz, new_xyz, offset, new_offset, k):
    m = new_xyz.shape[0]
    idx = np.zeros((m, k), np.int32); d2 = np.zeros((m, k), np.float32)
    lib.ref_knnquery(m, k, P(xyz), P(new_xyz), P(offset), P(new_offset), P(idx), P(d2))
    return idx, d2


def fps(lib, xyz, offset, new_offset):
    n = xyz.shape[0]
    n_max = int(max(np.diff(np.concatenate([[0], offset]))))
    tmp = np.full((n,), 1e10, np.float32)
    idx = np.zeros((int(new_offset[-1]),), np.int32)
    lib.ref_furthestsampling(len(offset), n_max, P(xyz), P(offset), P(new_offset), P(tmp), P(idx))
    return idx, n_max, tmp


def lattice(n_side, scale=1.0):
    g = np.arange(n_side, dtype=np.float32) * np.float32(scale)
    return f32(np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3))


def main():
    assert os.path.isdir(REF), "needs /root/reference (build container only)"
    with tempfile.TemporaryDirectory() as tmp:
        lib = build_ref(tmp)
        out = {}

        # ---------------- K1 knnquery ----------------
        cases = {}
        rng = np.random.default_rng(0)
        # C1: S-uniform(4096, seed 0), K=16, self query
        pts = f32(rng.uniform(0, 1, (4096, 3)) * np.array([0.95, 0.95, 0.95]))
        cases["c1_uniform4096_k16"] = (pts, pts, i32([4096]), i32([4096]), 16)
        # 3 clouds, cumulative ends [100, 101, 612] (one 1-point cloud), queries = subset, K=8 (> n_b for cloud 1)
        rng = np.random.default_rng(1)
        pts3 = f32(rng.normal(size=(612, 3)))
        q3 = f32(np.concatenate([pts3[0:100:2], pts3[100:101], pts3[101:612:3]]) + rng.normal(size=(50 + 1 + 171, 3)).astype(np.float32) * 0.01)
        cases["three_clouds_k8"] = (pts3, q3, i32([100, 101, 612]), i32([50, 51, 222]), 8)
        # tie-heavy lattice 8^3, self query, several K
        lat = lattice(8)
        for k in (2, 5, 16, 27):
            cases[f"lattice8_k{k}"] = (lat, lat, i32([512]), i32([512]), k)
        # shuffled lattice (index order no longer spatial)
        perm = np.random.default_rng(2).permutation(512)
        cases["lattice8_shuffled_k16"] = (f32(lat[perm]), f32(lat[perm]), i32([512]), i32([512]), 16)
        # n_b < K: sentinel rows
        rng = np.random.default_rng(3)
        small = f32(rng.uniform(size=(7 + 3, 3)))
        cases["fewer_than_k"] = (small, small, i32([7, 10]), i32([7, 10]), 9)
        # SURVEY §7 known answer: query origin, 6 supports with d2 = 5,5,1,1,5,5
        ka = f32([[1, 2, 0], [2, 1, 0], [1, 0, 0], [0, 1, 0], [2, -1, 0], [-1, 2, 0]])
        for k in (2, 3, 4, 5):
            cases[f"survey_ka_k{k}"] = (ka, f32([[0, 0, 0]]), i32([6]), i32([1]), k)
        cases["survey_ka_n2_k4"] = (f32(ka[:2]), f32([[0, 0, 0]]), i32([2]), i32([1]), 4)
        # large K (CBL sub-scene label path), coarse queries vs fine supports
        rng = np.random.default_rng(4)
        fine = f32(rng.uniform(size=(3000, 3)))
        cases["subscene_k64"] = (fine, f32(fine[::50]), i32([1400, 3000]), i32([28, 60]), 64)
        cases["subscene_k256"] = (fine, f32(fine[::100]), i32([1400, 3000]), i32([14, 30]), 256)
        # duplicated points (zero distances tie)
        rng = np.random.default_rng(5)
        dup = f32(np.repeat(rng.uniform(size=(64, 3)), 4, axis=0))
        cases["duplicates_k6"] = (dup, dup, i32([256]), i32([256]), 6)
        for name, (xyz, q, off, noff, k) in cases.items():
            idx, d2 = knn(lib, xyz, q, off, noff, k)
            out[f"knn/{name}/xyz"] = xyz; out[f"knn/{name}/new_xyz"] = q
            out[f"knn/{name}/offset"] = off; out[f"knn/{name}/new_offset"] = noff
            out[f"knn/{name}/k"] = np.int32(k); out[f"knn/{name}/idx"] = idx; out[f"knn/{name}/dist2"] = d2
        np.savez_compressed(os.path.join(OUT, "pointops_knn.npz"), **out)
        print("knn cases:", len(cases))

        # ---------------- K2 furthest sampling ----------------
        out = {}
        cases = {}
        rng = np.random.default_rng(10)
        cases["random1000_to_250"] = (f32(rng.uniform(size=(1000, 3))), i32([1000]), i32([250]))
        cases["lattice6_to_54"] = (lattice(6), i32([216]), i32([54]))
        cases["lattice8_to_128"] = (lattice(8), i32([512]), i32([128]))
        rng = np.random.default_rng(11)
        cases["two_clouds_unequal"] = (f32(rng.normal(size=(300 + 77, 3))), i32([300, 377]), i32([75, 94]))
        rng = np.random.default_rng(12)
        cases["three_clouds_tiny"] = (f32(rng.normal(size=(40 + 1 + 9, 3))), i32([40, 41, 50]), i32([10, 11, 13]))
        rng = np.random.default_rng(13)
        cases["random2500_to_625_b1024"] = (f32(rng.uniform(size=(2500, 3))), i32([2500]), i32([625]))
        rng = np.random.default_rng(14)
        dup = f32(np.repeat(rng.uniform(size=(50, 3)), 4, axis=0))   # exhausts distinct points -> all-zero ties
        cases["duplicates200_to_100"] = (dup, i32([200]), i32([100]))
        for name, (xyz, off, noff) in cases.items():
            idx, n_max, tmp = fps(lib, xyz, off, noff)
            out[f"fps/{name}/xyz"] = xyz; out[f"fps/{name}/offset"] = off; out[f"fps/{name}/new_offset"] = noff
            out[f"fps/{name}/n_max"] = np.int32(n_max); out[f"fps/{name}/idx"] = idx; out[f"fps/{name}/tmp_after"] = tmp
            out[f"fps/{name}/block"] = np.int32(lib.ref_opt_n_threads(n_max))
        out["opt_n_threads/n"] = np.arange(1, 5000, dtype=np.int32)
        out["opt_n_threads/threads"] = i32([lib.ref_opt_n_threads(int(n)) for n in range(1, 5000)])
        np.savez_compressed(os.path.join(OUT, "pointops_fps.npz"), **out)
        print("fps cases:", len(cases))

        # ---------------- K3..K10 ----------------
        out = {}
        rng = np.random.default_rng(20)
        n, m, ns, c, wc, k = 97, 61, 5, 12, 4, 3
        inp = f32(rng.normal(size=(n, c)))
        idx = i32(rng.integers(0, n, size=(m, ns)))
        o = np.empty((m, ns, c), np.float32)
        lib.ref_grouping_forward(m, ns, c, P(inp), P(idx), P(o))
        go = f32(rng.normal(size=(m, ns, c))); gi = np.zeros((n, c), np.float32)
        lib.ref_grouping_backward(m, ns, c, P(go), P(idx), P(gi))
        out.update({"grouping/input": inp, "grouping/idx": idx, "grouping/output": o, "grouping/grad_output": go, "grouping/grad_input": gi})

        src = f32(rng.normal(size=(m, c))); iidx = i32(rng.integers(0, m, size=(n, k)))
        w = f32(rng.uniform(size=(n, k))); w /= w.sum(1, keepdims=True); w = f32(w)
        io = np.zeros((n, c), np.float32)
        lib.ref_interpolation_forward(n, c, k, P(src), P(iidx), P(w), P(io))
        igo = f32(rng.normal(size=(n, c))); igi = np.zeros((m, c), np.float32)
        lib.ref_interpolation_backward(n, c, k, P(igo), P(iidx), P(w), P(igi))
        out.update({"interpolation/input": src, "interpolation/idx": iidx, "interpolation/weight": w, "interpolation/output": io,
                    "interpolation/grad_output": igo, "interpolation/grad_input": igi})

        a = f32(rng.normal(size=(n, c))); bb = f32(rng.normal(size=(n, c))); sidx = i32(rng.integers(0, n, size=(n, ns)))
        so_ = np.zeros((n, ns, c), np.float32)
        lib.ref_subtraction_forward(n, ns, c, P(a), P(bb), P(sidx), P(so_))
        sgo = f32(rng.normal(size=(n, ns, c))); g1 = np.zeros((n, c), np.float32); g2 = np.zeros((n, c), np.float32)
        lib.ref_subtraction_backward(n, ns, c, P(sidx), P(sgo), P(g1), P(g2))
        out.update({"subtraction/input1": a, "subtraction/input2": bb, "subtraction/idx": sidx, "subtraction/output": so_,
                    "subtraction/grad_output": sgo, "subtraction/grad_input1": g1, "subtraction/grad_input2": g2})

        pos = f32(rng.normal(size=(n, ns, c))); ww = f32(rng.normal(size=(n, ns, wc)))
        ao = np.zeros((n, c), np.float32)
        lib.ref_aggregation_forward(n, ns, c, wc, P(a), P(pos), P(ww), P(sidx), P(ao))
        ago = f32(rng.normal(size=(n, c)))
        agi = np.zeros((n, c), np.float32); agp = np.zeros((n, ns, c), np.float32); agw = np.zeros((n, ns, wc), np.float32)
        lib.ref_aggregation_backward(n, ns, c, wc, P(a), P(pos), P(ww), P(sidx), P(ago), P(agi), P(agp), P(agw))
        out.update({"aggregation/input": a, "aggregation/position": pos, "aggregation/weight": ww, "aggregation/idx": sidx,
                    "aggregation/output": ao, "aggregation/grad_output": ago, "aggregation/grad_input": agi,
                    "aggregation/grad_position": agp, "aggregation/grad_weight": agw})
        np.savez_compressed(os.path.join(OUT, "pointops_k3_k10.npz"), **out)
        print("k3..k10 written")


if __name__ == "__main__":
    sys.exit(main())
