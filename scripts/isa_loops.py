"""Instruction mix of every long loop in the kernels' ISA (rnamc_kernels.s)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "rna_algos_amd/csrc/rnamc_kernels.s"
lines = open(path).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN5rnamc\S+: ", l)]
starts.append((len(lines), "end"))
for (a0, name), (a1, _) in zip(starts, starts[1:]):
    if "k_inside" not in name and "k_outside" not in name:
        continue
    body = lines[a0:a1]
    labels = {}
    for idx, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = idx
    print(name[-45:])
    for idx, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < idx and idx - labels[m.group(1)] > 150:
            a, b = labels[m.group(1)], idx
            bb = body[a:b + 1]
            valu = sum(1 for x in bb if re.match(r"\s+v_", x))
            salu = sum(1 for x in bb if re.match(r"\s+s_", x) and "s_waitcnt" not in x and "s_nop" not in x)
            vmem = sum(1 for x in bb if "global_load" in x or "global_store" in x)
            lds = sum(1 for x in bb if re.match(r"\s+ds_", x))
            br = sum(1 for x in bb if "s_cbranch" in x)
            wt = sum(1 for x in bb if "s_waitcnt" in x)
            print(f"  loop {a}-{b} ({b - a} lines): valu={valu} salu={salu} vmem={vmem} lds={lds} "
                  f"branches={br} waits={wt}")
