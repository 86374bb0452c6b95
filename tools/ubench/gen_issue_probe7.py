#!/usr/bin/env python3
"""Generates tools/ubench/issue_probe7.hip -- fourth round (see gen_issue_probe*.py).

Rounds 1-3 timed INDEPENDENT instructions (eight registers in turn).  The bulk kernel's fast-class instructions are mostly
links of dependent chains (sub -> lshr -> and ...), and dropped from the replayed loop they turn out to cost 3.7-4.6 cycles
each, not 2.1-2.6 (profiles/r08_replay_drop.txt).  This round: the same fast-class instruction with 1, 2, 4, 8 independent
chains per wave (k registers in turn: every instruction depends on the one k places before it), at 1-8 waves per SIMD, and
the dependent chain broken up by separators or by slow-class instructions of another chain.
"""
import os

from gen_issue_probe import HEAD, kernel

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = {
    "add": "v_add_u32 %{r}, %{r}, %8",
    "xor": "v_xor_b32 %{r}, %{r}, %8",
    "lshr": "v_lshrrev_b32 %{r}, 1, %{r}",
    "bitop3": "v_bitop3_b32 %{r}, %{r}, %8, %8 bitop3:0x96",
    "sel64": "v_cndmask_b32_e64 %{r}, %{r}, %8, s[30:31]",
    "bcnt": "v_bcnt_u32_b32 %{r}, %{r}, %8",
    "mad64": "v_mad_u64_u32 v[44:45], s[28:29], %{r}, %8, v[44:45]",
    "mad64d": "v_mad_u64_u32 v[{p}:{q}], s[28:29], v{p}, %8, v[{p}:{q}]",
    "mad64i": "v_mad_u64_u32 v[{p}:{q}], s[28:29], %8, %8, v[60:61]",
    "mulhi": "v_mul_hi_u32 %{r}, %{r}, %8",
    "mullo": "v_mul_lo_u32 %{r}, %{r}, %8",
}


CL = ["v40", "v42"] + ["v%d" % r for r in range(44, 62)] + ["s28", "s29", "vcc"]


def chain(op, k, total=2048, sep_every=0):
    lines = []
    for i in range(total):
        lines.append(OPS[op].format(r=i % k, p=44 + 2 * (i % k), q=45 + 2 * (i % k)))
        if sep_every and (i + 1) % sep_every == 0:
            lines.append("s_nop 0")
    return "\n".join(lines), total


def main():
    out = [HEAD]
    table = []
    n = 0
    for op in ("mad64d", "mad64i", "mulhi", "mullo", "bcnt"):
        for k in (1, 2, 4):
            body, nv = chain(op, k)
            name = "k_%d" % n
            n += 1
            out.append(kernel(name, "v_mov_b32 v42, 0\n" + body, 1, CL))
            table.append(("%s, %d independent chain(s) per wave" % (op, k), name, nv))
    for k, se in ((1, 1), (2, 1), (4, 1)):
        body, nv = chain("mad64d", k, sep_every=se)
        name = "k_%d" % n
        n += 1
        out.append(kernel(name, "v_mov_b32 v42, 0\n" + body, 1, CL))
        table.append(("mad64d, %d chain(s), s_nop 0 behind every %d" % (k, se), name, nv))
    out.append("struct B { const char *name; void (*k)(uint32_t *, int); int nv; };\n")
    out.append("static const B bs[] = {\n" + "".join('    {"%s", %s, %d},\n' % t for t in table) + "};\n")
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
    printf("SIMD-cycles per VALU instruction (2.4 GHz assumed) by waves per SIMD:          1      2      4      8\n");
    for (const B &b : bs) {
        printf("%-52s", b.name);
        for (int W : {1, 2, 4, 8}) {
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 64, 0, out, 5);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 64, 0, out, iters);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %6.2f", ms * 1e-3 * ghz * 1e9 / ((double)iters * b.nv * W));
        }
        printf("\n");
    }
    return 0;
}
''')
    with open(os.path.join(HERE, "issue_probe7.hip"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
