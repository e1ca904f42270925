"""Generator of tools/ubench/bin/mfma_fill.hip -- what hides in the gaps between v_mfma_f32_32x32x16_bf16 on ONE SIMD of gfx950?

    python tools/ubench/gen_mfma_fill.py && hipcc --offload-arch=gfx950 -O3 tools/ubench/bin/mfma_fill.hip -o tools/ubench/bin/mfma_fill
    tools/ubench/bin/mfma_fill > profiles/r03_ubench_mfma_fill.txt        (on the GPU box)

Every loop body is ONE asm volatile block, so the placement below is exactly what the SIMD sees (no compiler scheduling, no
hazard nops: the fillers never touch an accumulator).  A body is 8 MFMAs (alternating two accumulators, or one dependent
chain) and 8 F filler instructions on 8..16 INDEPENDENT registers, placed either
    I  interleaved: MFMA, F fillers, MFMA, F fillers, ...            (the hand-placed stream of MI355X_MICROARCH.md:447)
    B  block      : 8 MFMAs, then 8 F fillers                         (what an un-pipelined epilogue looks like)
at 1, 2 and 4 waves per SIMD, plus the role split  R: half of the waves of a SIMD issue only MFMAs, the other half only the
fillers.  Timed with s_memtime inside the kernel (shader cycles, independent of DVFS); printed as SIMD cycles per MFMA.
"""
import os

FILLERS = {
    # name: (asm template with {d} = destination register operand index, {s} = source), reads/writes independent registers
    "v_fma_f32": "v_fma_f32 %{d}, %{d}, %[m], %[c]",
    "v_max_i32": "v_max_i32 %{d}, 0, %{d}",
    "v_cvt_pk_bf16_f32": "v_cvt_pk_bf16_f32 %{d}, %{d}, %[m]",
    "v_dot2c_f32_bf16": "v_dot2c_f32_bf16 %{d}, %[m], %[c]",
    "v_sub_f32": "v_sub_f32 %{d}, %{d}, %[m]",
    "v_exp_f32": "v_exp_f32 %{d}, %{d}",
    "ds_read_b128": None,      # handled separately (LDS class, own issue port)
}
NREG = 16


def body(op, F, placement, chain):
    """asm text of one loop body: 8 MFMAs + 8 F fillers"""
    lines = []
    reg = [0]

    def filler():
        if op == "ds_read_b128":
            r = reg[0] % 4
            reg[0] += 1
            return "ds_read_b128 %%[q%d], %%[la] offset:%d" % (r, 1024 * (reg[0] % 16))
        t = FILLERS[op].replace("{d}", "[r%d]" % (reg[0] % NREG))
        reg[0] += 1
        return t.replace("%[", "%[")

    def mfma(i):
        a = "acc0" if (chain or i % 2 == 0) else "acc1"
        return "v_mfma_f32_32x32x16_bf16 %%[%s], %%[a], %%[b], %%[%s]" % (a, a)

    if placement == "I":
        for i in range(8):
            lines.append(mfma(i))
            for _ in range(F):
                lines.append(filler())
    elif placement == "B":
        for i in range(8):
            lines.append(mfma(i))
        for _ in range(8 * F):
            lines.append(filler())
    elif placement == "M":      # MFMAs only
        for i in range(8):
            lines.append(mfma(i))
    elif placement == "V":      # fillers only
        for _ in range(8 * F):
            lines.append(filler())
    if op == "ds_read_b128" and placement != "M":
        lines.append("s_waitcnt lgkmcnt(0)")
    return lines


def kernel(name, op, F, placement, chain, split=False):
    def asm_block(lines):
        txt = "\\n\\t".join(l.replace("%%", "%") for l in lines)
        outs = ', '.join(['[acc0] "+v"(acc0)', '[acc1] "+v"(acc1)'] + ['[r%d] "+v"(r[%d])' % (i, i) for i in range(NREG)] +
                         ['[q%d] "=&v"(q[%d])' % (i, i) for i in range(4)])
        ins = '[a] "v"(a), [b] "v"(b), [m] "v"(m), [c] "v"(c), [la] "v"(la)'
        return '            asm volatile("%s"\n                         : %s\n                         : %s);\n' % (txt, outs, ins)

    s = "__global__ __launch_bounds__(1024) void %s(unsigned long long* cyc, float* sink, int iters, int nmf) {\n" % name
    s += "    __shared__ __attribute__((aligned(16))) float lds[16 * 1024 / 4 + 64 * 4];\n"
    s += "    PROLOGUE\n"
    if split:
        s += "    const bool mf = (int)(threadIdx.x >> 6) < nmf;\n"
        s += "    const unsigned long long t0 = clock64();\n"
        s += "    if (mf) {\n        for (int it = 0; it < iters; ++it) {\n" + asm_block(body(op, F, "M", chain)) + "        }\n    } else {\n"
        s += "        for (int it = 0; it < iters; ++it) {\n" + asm_block(body(op, F, "V", chain)) + "        }\n    }\n"
    else:
        s += "    const unsigned long long t0 = clock64();\n"
        s += "    for (int it = 0; it < iters; ++it) {\n" + asm_block(body(op, F, placement, chain)) + "    }\n"
    s += "    EPILOGUE\n}\n\n"
    return s


def main():
    root = os.path.dirname(os.path.abspath(__file__))
    os.makedirs(os.path.join(root, "bin"), exist_ok=True)
    ks, table = [], []
    for op in ("v_fma_f32", "v_max_i32", "v_cvt_pk_bf16_f32", "v_dot2c_f32_bf16", "v_exp_f32", "ds_read_b128"):
        fs = (0, 2, 4, 5, 6, 8, 12) if op in ("v_fma_f32", "v_max_i32") else ((1, 2) if op == "ds_read_b128" else (4, 6, 8))
        for F in fs:
            for pl in ("I", "B"):
                if F == 0 and pl == "B":
                    continue
                for chain in ((0, 1) if op == "v_fma_f32" and F in (0, 5, 8) else (0,)):
                    nm = "k_%s_F%d_%s_c%d" % (op, F, pl, chain)
                    ks.append(kernel(nm, op, F, pl, chain))
                    table.append((nm, op, F, pl, chain, 0))
        for F in ((4, 8) if op != "ds_read_b128" else (2,)):
            nm = "k_%s_F%d_R" % (op, F)
            ks.append(kernel(nm, op, F, "R", 0, split=True))
            table.append((nm, op, F, "R", 0, 1))
    src = r'''// GENERATED by tools/ubench/gen_mfma_fill.py -- do not edit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define PROLOGUE \
    for (int e = threadIdx.x; e < 16 * 1024 / 4 + 64 * 4; e += blockDim.x) lds[e] = (float)e; \
    __syncthreads(); \
    bf16x8 a, b; \
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); } \
    f32x16 acc0 = {}, acc1 = {}; \
    float r[16]; \
    for (int j = 0; j < 16; ++j) r[j] = threadIdx.x * 0.25f + j; \
    f32x4 q[4] = {}; \
    float m = 1.0001f, c = 0.5f; \
    asm volatile("" : "+v"(m), "+v"(c)); \
    unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16;
#define EPILOGUE \
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); \
    const unsigned long long t1 = clock64(); \
    float s = 0.0f; \
    for (int j = 0; j < 16; ++j) s += r[j] + acc0[j] + acc1[j]; \
    for (int j = 0; j < 4; ++j) s += q[j][0] + q[j][3]; \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s; \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;

''' + "".join(ks)
    src += "struct Row { const char* name; const char* op; int F; char pl; int chain; int split; void (*fn)(unsigned long long*, float*, int, int); };\n"
    src += "static Row rows[] = {\n" + "".join('    {"%s", "%s", %d, \'%s\', %d, %d, %s},\n' % (t[0], t[1], t[2], t[3], t[4], t[5], t[0]) for t in table) + "};\n"
    src += r'''
int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;
    unsigned long long* cyc; float* sink;
    (void)hipMalloc(&cyc, 256 * 16 * 8); (void)hipMalloc(&sink, 256 * 1024 * 4);
    std::vector<unsigned long long> h(256 * 16);
    const int iters = 4000;
    printf("# SIMD cycles per MFMA (s_memtime), body = 8 MFMA 32x32x16 bf16 + 8 F fillers; W = waves per SIMD; 256 work-groups\n");
    printf("# I interleaved (MFMA, F fillers, ...), B block (8 MFMA then 8 F fillers), R role split (half the waves MFMA-only, half filler-only)\n");
    printf("%-22s %3s %2s %5s | %9s %9s %9s\n", "filler", "F", "pl", "chain", "W=1", "W=2", "W=4");
    for (const Row& r : rows) {
        if (only && !strstr(r.name, only)) continue;
        printf("%-22s %3d %2c %5d |", r.op, r.F, r.pl, r.chain);
        for (int W : {1, 2, 4}) {
            if (r.split && W == 1) { printf(" %9s", "-"); continue; }
            const int nw = 4 * W;                                   // waves per work-group (one work-group per CU)
            // role split: the first half of a SIMD's waves (waves w with (w / 4) < W / 2 -- waves go to SIMDs round robin) issue MFMAs
            const int nmf = r.split ? nw / 2 : 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(r.fn, dim3(256), dim3(64 * nw), 0, 0, cyc, sink, iters, nmf);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h.data(), cyc, 256 * nw * 8, hipMemcpyDeviceToHost);
            double sm = 0, sv = 0; int nm = 0, nv = 0;
            for (int bq = 0; bq < 256; ++bq) for (int w = 0; w < nw; ++w) {
                const double v = (double)h[bq * nw + w];
                if (r.split && w >= nmf) { sv += v; ++nv; } else { sm += v; ++nm; }
            }
            // MFMAs issued on one SIMD during the kernel: (MFMA-issuing waves per SIMD) x iters x 8
            const double mw = r.split ? W / 2.0 : (double)W;
            const double per = (sm / nm) / (mw * iters * 8.0);
            if (r.split) printf(" %4.1f/%4.1f", per, (sv / nv) / ((W / 2.0) * iters * 8.0));
            else printf(" %9.1f", per);
        }
        printf("\n");
    }
    return 0;
}
'''
    out = os.path.join(root, "bin", "mfma_fill.hip")
    open(out, "w").write(src)
    print(out, len(table), "kernels")


if __name__ == "__main__":
    main()
