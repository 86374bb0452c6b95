#!/usr/bin/env python3
"""Generates tools/ubench/issue_probe6.hip -- sixth round: MIXED fast / slow streams whose instructions DEPEND on each other
(k registers in turn: every instruction reads the result k places before it), with and without separators.  Rounds 1-3 had
eight independent chains; the kernels' chains are `perm -> sub -> lshr -> and -> sad_u8 -> add3`: one."""
import os

from gen_issue_probe import HEAD, kernel

HERE = os.path.dirname(os.path.abspath(__file__))
T = {"f": "v_add_u32 %{r}, %{r}, %8", "g": "v_and_b32 %{r}, %{r}, %8", "S": "v_bcnt_u32_b32 %{r}, %{r}, %8",
     "P": "v_perm_b32 %{r}, %{r}, %8, %8", "A": "v_add3_u32 %{r}, %{r}, %8, %8", "n": "s_nop 0"}
PATTERNS = ["f", "S", "fS", "fSn", "ffS", "ffSn", "fffS", "fffSn", "fffSSn", "fffSnSn", "PfgfSA", "PnfgfSnAn", "PfgfSAn"]


def expand(pattern, k, total=1536):
    lines, i, nv, nf, ns = [], 0, 0, 0, 0
    while nv < total:
        for ch in pattern:
            if ch == "n":
                lines.append(T[ch])
                continue
            lines.append(T[ch].format(r=i % k))
            nv += 1
            if ch in "fg":
                nf += 1
            else:
                ns += 1
        i += 1  # the whole pattern on one register: a chain; the next repetition on the next of k registers
    return "\n".join(lines), nv, nf, ns


def main():
    out = [HEAD]
    table = []
    n = 0
    for pat in PATTERNS:
        for k in (1, 2, 4, 8):
            body, nv, nf, ns = expand(pat, k)
            name = "k_%d" % n
            n += 1
            out.append(kernel(name, body, 1, ["vcc"]))
            table.append(("%-10s %d chain(s)" % (pat, k), name, nv, nf, ns))
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
    printf("f = v_add_u32, g = v_and_b32, S = v_bcnt, P = v_perm_b32, A = v_add3_u32, n = s_nop 0; a pattern runs on ONE register (a\n"
           "dependent chain), its repetitions on k registers in turn.  SIMD-cycles per VALU instruction at 4 waves per SIMD | additive (2.1 / 4.1)\n");
    for (const B &b : bs) {
        hipLaunchKernelGGL(b.k, dim3(n_cu * 4), dim3(256), 64, 0, out, 5);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(b.k, dim3(n_cu * 4), dim3(256), 64, 0, out, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %6.2f   %5.2f\n", b.name, ms * 1e-3 * ghz * 1e9 / ((double)iters * b.nv * 4), (2.1 * b.nf + 4.1 * b.ns) / b.nv);
    }
    return 0;
}
''')
    with open(os.path.join(HERE, "issue_probe6.hip"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
