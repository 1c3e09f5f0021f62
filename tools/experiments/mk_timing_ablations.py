"""Regenerates tools/experiments/timing_ablations.patch against the CURRENT product sources: the wrong-value timing ablations are written
here as (anchor -> replacement) pairs, applied to a temporary copy of csrc/, and diffed.  Run after editing the kernels:
    python3 tools/experiments/mk_timing_ablations.py"""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "csrc")
TMP = tempfile.mkdtemp()
for f in ("otmb_tm_column.h", "otmb_transportmatrix.hip"):
    shutil.copy(os.path.join(CSRC, f), TMP)
def edit(path, pairs):
    s=open(path,encoding='utf-8').read()
    for old,new in pairs:
        assert s.count(old)==1,(path,old[:70],s.count(old))
        s=s.replace(old,new)
    open(path,'w',encoding='utf-8').write(s)
edit(os.path.join(TMP, 'otmb_tm_column.h'),[
("#define FDIV(a, b) ((a) / (b))\n","""#ifdef OTMB_DBG_MULDIV  // timing experiment only: what the 28 divisions of a column cost (wrong values)
#define FDIV(a, b) ((a) * (b))
#else
#define FDIV(a, b) ((a) / (b))
#endif
"""),
("__device__ __forceinline__ double ldv(const char *b, unsigned byteoff) { return ldd(b, byteoff); }\n","""#ifdef OTMB_DBG_NOVALLOAD  // timing experiment only (wrong values): value-only inputs are not loaded, the pattern inputs are
__device__ __forceinline__ double ldv(const char *b, unsigned byteoff) { return 1.0 + (double)byteoff * 1e-9 + (double)((size_t)b & 0xfff) * 1e-7; }
#else
__device__ __forceinline__ double ldv(const char *b, unsigned byteoff) { return ldd(b, byteoff); }
#endif
"""),
("""    const double vC = ldv(tb.v, oC), vE = ldv(tb.v, oE), vW = ldv(tb.v, oW), vS = ldv(tb.v, oS), vN = ldv(tb.v, oN),
                 vA = ldv(tb.v, oA), vB = ldv(tb.v, oB);
""","""#ifdef OTMB_DBG_NOEW  // timing experiment only (wrong values): no second load instruction into lines that are in flight
    const double vC = ldv(tb.v, oC), vE = vC, vW = vC, vS = ldv(tb.v, oS), vN = ldv(tb.v, oN), vA = ldv(tb.v, oA), vB = ldv(tb.v, oB);
#else
    const double vC = ldv(tb.v, oC), vE = ldv(tb.v, oE), vW = ldv(tb.v, oW), vS = ldv(tb.v, oS), vN = ldv(tb.v, oN),
                 vA = ldv(tb.v, oA), vB = ldv(tb.v, oB);
#endif
"""),
("        rE = ldv(tb.rho, oE); rW = ldv(tb.rho, oW);\n","""#ifdef OTMB_DBG_NOEW
        rE = rC; rW = rC;
#else
        rE = ldv(tb.rho, oE); rW = ldv(tb.rho, oW);
#endif
"""),
("""        tC = ldv(tb.thk, oC); tE = ldv(tb.thk, oE); tW = ldv(tb.thk, oW); tS = ldv(tb.thk, oS);
        tN = ldv(tb.thk, oN);
""","""#ifdef OTMB_DBG_NOEW
        tC = ldv(tb.thk, oC); tE = tC; tW = tC; tS = ldv(tb.thk, oS); tN = ldv(tb.thk, oN);
#else
        tC = ldv(tb.thk, oC); tE = ldv(tb.thk, oE); tW = ldv(tb.thk, oW); tS = ldv(tb.thk, oS);
        tN = ldv(tb.thk, oN);
#endif
"""),
("""        eE_w = ldv(eEp, sW); dE_w = ldv(dEp, sW);  // west cell's east edge / distance to its east nbr
        eW_e = ldv(eWp, sE); dW_e = ldv(dWp, sE);
""","""#ifdef OTMB_DBG_NOEW
        eE_w = eE_c; dE_w = dE_c; eW_e = eW_c; dW_e = dW_c;
#else
        eE_w = ldv(eEp, sW); dE_w = ldv(dEp, sW);  // west cell's east edge / distance to its east nbr
        eW_e = ldv(eWp, sE); dW_e = ldv(dWp, sE);
#endif
"""),
])
edit(os.path.join(TMP, 'otmb_transportmatrix.hip'),[
("    if (p.next_state && blockIdx.x == 0 && tid < (int)(OTMB_TM_STATE_BYTES / sizeof(int))) p.next_state[tid] = 0;\n","""    if (p.next_state && blockIdx.x == 0 && tid < (int)(OTMB_TM_STATE_BYTES / sizeof(int))) p.next_state[tid] = 0;
#ifdef OTMB_STAGGER_UNITS
    // Experiment: the workgroups of the first dispatch round start together and march through their phases (loads,
    // arithmetic, stores) in lockstep; delay the k-th workgroup of a CU by k * OTMB_STAGGER_UNITS * 64 * 127 cycles
    if (blockIdx.x < 1024) {
        const int slot = blockIdx.x / 256;
        for (int q = 0; q < slot * OTMB_STAGGER_UNITS; ++q) __builtin_amdgcn_s_sleep(127);
    }
#endif
"""),
("""                else {
                    canonical = ldi(tb.lw, oC) == c;
                    if (canonical) {
""","""#ifdef OTMB_DBG_NOGENERIC  // timing experiment only (wrong on the seam row)
                else if (true) { canonical = true; col.padv = col.phh = col.pml = col.pdp = 0; }
#endif
                else {
                    canonical = ldi(tb.lw, oC) == c;
                    if (canonical) {
"""),
("""        if (live) {
            const unsigned q0 = ex[m] - wb[m];""","""#ifdef OTMB_DBG_NOLDS
        if (false) {
#else
        if (live) {
#endif
            const unsigned q0 = ex[m] - wb[m];"""),
("""        if (room) {
            const unsigned end = cnt;""","""#ifdef OTMB_DBG_NOSTORE
        room = false;
#endif
        if (room) {
            const unsigned end = cnt;"""),
("""            for (unsigned base = 0; base < end; base += 128) {  // full pairs
                const unsigned u = base + 2 * lane;
                if (u + 1 < end) {
                    TM_STORE(*(const i64x2 *)(my_row + u), (i64x2g *)(rvb + u * 8u));
                    TM_STORE(*(const i64x2 *)(my_val + u), (i64x2g *)(nzb + u * 8u));
                }
            }
""","""#ifdef OTMB_STORE_SPLIT  // experiment (profiles/r05): all rowval pieces of the run, then all nzval pieces -- one output stream per burst
            for (unsigned base = 0; base < end; base += 128) {
                const unsigned u = base + 2 * lane;
                if (u + 1 < end) TM_STORE(*(const i64x2 *)(my_row + u), (i64x2g *)(rvb + u * 8u));
            }
            for (unsigned base = 0; base < end; base += 128) {
                const unsigned u = base + 2 * lane;
                if (u + 1 < end) TM_STORE(*(const i64x2 *)(my_val + u), (i64x2g *)(nzb + u * 8u));
            }
#else
            for (unsigned base = 0; base < end; base += 128) {  // full pairs
                const unsigned u = base + 2 * lane;
                if (u + 1 < end) {
                    TM_STORE(*(const i64x2 *)(my_row + u), (i64x2g *)(rvb + u * 8u));
                    TM_STORE(*(const i64x2 *)(my_val + u), (i64x2g *)(nzb + u * 8u));
                }
            }
#endif
"""),
])

HEADER = """# Timing ablations that compute WRONG values (or wrong tile assignments), kept OUT of the product sources since round 5:
#   OTMB_DBG_MULDIV, _NOVALLOAD, _NOEW, _NOGENERIC, _NOLDS, _NOSTORE, OTMB_STAGGER_UNITS, and the store-order experiment OTMB_STORE_SPLIT.
# Apply from oceantransportmatrixbuilder.jl_amd/csrc/ (patch -p0 < ../../tools/experiments/timing_ablations.patch), build a VARIANT library with the -D flag
# (build.build(name=..., extra=[...])), run tools/gpu_ab.sh with LIB=<name>, and revert.  tests/test_sources_clean.py refuses these macros in csrc/.
# (Regenerate after kernel edits: python3 tools/experiments/mk_timing_ablations.py)
"""
out = HEADER
for f in ("otmb_tm_column.h", "otmb_transportmatrix.hip"):
    r = subprocess.run(["diff", "-u", "--label", f, "--label", f, f, os.path.join(TMP, f)], cwd=CSRC, capture_output=True, text=True)
    out += r.stdout
open(os.path.join(ROOT, "tools", "experiments", "timing_ablations.patch"), "w", encoding="utf-8").write(out)
shutil.rmtree(TMP)
print("tools/experiments/timing_ablations.patch regenerated")
