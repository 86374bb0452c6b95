#!/usr/bin/env python3
"""Generates tools/ubench/issue_probe3.hip -- third round (see gen_issue_probe.py, gen_issue_probe2.py).

issue_probe2 found that a NON-VALU instruction (s_nop, s_waitcnt, a branch, s_setprio, s_mov exec ...) directly behind a
4-cycle-class instruction ends the slow cadence it starts: 127 adds, v_bcnt, 128 adds costs 3.66 cycles per instruction,
with an s_nop 0 behind the v_bcnt 2.20.  This round: dense mixes as the kernels have them (every second or third
instruction of the slow class) with and without such a separator, which instructions separate, and where it must sit.
f = v_add_u32 (fast class), S = v_bcnt_u32_b32 (slow class), n = s_nop 0.
"""
import os

from gen_issue_probe import HEAD, kernel

HERE = os.path.dirname(os.path.abspath(__file__))

F = "v_add_u32 %{r}, %{r}, %8"
S = "v_bcnt_u32_b32 %{r}, %{r}, %8"
SEP = {
    "n": "s_nop 0",
    "w": "s_waitcnt lgkmcnt(0)",
    "m": "s_mov_b32 s20, s21",
    "a": "s_and_b64 s[22:23], s[24:25], s[26:27]",
    "d": "ds_read_b32 v40, v42",
    "p": "s_setprio 0",
    "N": "s_nop 3",
    "v": "v_nop",
    "M": "v_mad_u64_u32 v[44:45], s[28:29], %{r}, %8, v[44:45]",
    "C": "v_cmp_eq_u32 vcc, %{r}, %8",
    "X": "v_max_u32 %{r}, %{r}, %8",
    "L": "v_lshl_add_u32 %{r}, %{r}, 4, %8",
    "B": "v_bitop3_b32 %{r}, %{r}, %8, %8 bitop3:0xe0",
    "K": "v_cndmask_b32_e64 %{r}, %{r}, %8, s[30:31]",
}


def expand(pattern, total=2048):
    """pattern string -> asm text of ~total VALU instructions, (n_valu, n_fast, n_slow)"""
    lines, r, nv, nf, ns = [], 0, 0, 0, 0
    while nv < total:
        for ch in pattern:
            if ch == "f":
                lines.append(F.format(r=r % 8)); r += 1; nv += 1; nf += 1
            elif ch == "S":
                lines.append(S.format(r=r % 8)); r += 1; nv += 1; ns += 1
            elif ch in "MCXL":
                lines.append(SEP[ch].format(r=r % 8)); r += 1; nv += 1; ns += 1
            elif ch in "BK":
                lines.append(SEP[ch].format(r=r % 8)); r += 1; nv += 1; nf += 1
            elif ch == "v":
                lines.append("v_nop"); nv += 1; nf += 1
            else:
                lines.append(SEP[ch])
    return "\n".join(lines), nv, nf, ns


PATTERNS = [
    "f", "S",
    "fS", "fSn", "fnS", "fnSn",
    "ffS", "ffSn", "fffS", "fffSn", "fffffffS", "fffffffSn",
    "fSS", "fSSn", "fSnSn", "ffSS", "ffSSn", "ffffSSSS", "ffffSSSSn", "ffffSnSnSnSn",
    "ffffffffSSSSSSSS", "ffffffffSSSSSSSSn", "ffffffffSSSSSSSSnn", "ffffffffSSSSSSSSN",
    "fSw", "fSm", "fSa", "fSd", "fSp", "fSN", "fSv",
    "ffffSSSSw", "ffffSSSSm", "ffffSSSSd",
    "fM", "fMn", "fC", "fCn", "fX", "fXn", "fL", "fLn",
    "fB", "fK", "BK", "fBKS", "fBKSn",
    "ffffMCXLn", "ffffMCXL",
    "fffSnfffSnfffSnfffSnfffSnfffSnfffSnfffS", "SnfffffffffffffffSnfffffffffffffff",
]


def main():
    out = [HEAD]
    table = []
    for k, pat in enumerate(PATTERNS):
        body, nv, nf, ns = expand(pat)
        name = "k_%d" % k
        out.append(kernel(name, "v_mov_b32 v42, 0\n" + body, 1,
                          ["v40", "v42", "v44", "v45", "s20", "s21", "s22", "s23", "s28", "s29", "vcc"]))
        table.append((pat, name, nv, nf, ns))
    out.append("struct B { const char *name; void (*k)(uint32_t *, int); int nv, nf, ns; };\n")
    out.append("static const B bs[] = {\n" + "".join('    {"%s", %s, %d, %d, %d},\n' % t for t in table) + "};\n")
    out.append(r'''
int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount, iters = 200;
    const double ghz = 2.4;
    uint32_t *out;
    CHK(hipMalloc(&out, (size_t)n_cu * 8 * 1024 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("f = v_add_u32, S = v_bcnt, n = s_nop 0, N = s_nop 3, w = s_waitcnt, m = s_mov_b32, a = s_and_b64, d = ds_read_b32, p = s_setprio,\n"
           "v = v_nop, M = v_mad_u64_u32, C = v_cmp -> vcc, X = v_max_u32, L = v_lshl_add_u32, B = v_bitop3_b32, K = v_cndmask_b32_e64 (sgpr mask)\n"
           "additive = (2.1 x fast + 4.1 x slow) / VALU\n");
    for (int W : {4}) {
        printf("---- %d waves per SIMD: pattern | SIMD-cycles per VALU instruction | additive model\n", W);
        for (const B &b : bs) {
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 64, 0, out, 5);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 64, 0, out, iters);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-44s %6.2f   %5.2f\n", b.name, ms * 1e-3 * ghz * 1e9 / ((double)iters * b.nv * W),
                   (2.1 * b.nf + 4.1 * b.ns) / b.nv);
        }
    }
    return 0;
}
''')
    with open(os.path.join(HERE, "issue_probe3.hip"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
