"""Per-kernel roofline table (markdown) of the bench step from a scripts/prof_decode.py breakdown (rocprofv3 durations)
and the algorithmic work of Llama-3.2-1B at 32 x 512 + 128.  usage: profiles_table.py <breakdown.txt>"""
import re, sys
M, H, F, V, NQKV = 16384, 2048, 8192, 128256, 3072
work = {  # (phase, site) -> (flops, bytes, bound)
    ("prefill", "ffn_up"): (2.0 * M * 2 * F * H, 0, "mfma"), ("prefill", "ffn_down"): (2.0 * M * H * F, 0, "mfma"),
    ("prefill", "qkv_proj"): (2.0 * M * NQKV * H, 0, "mfma"), ("prefill", "o_proj"): (2.0 * M * H * H, 0, "mfma"),
    ("prefill", "attention"): (34426847232.0, 0, "mfma"),
    ("decode", "ffn_up"): (0, 68288512, "hbm"), ("decode", "attention"): (0, 37781504, "hbm"),
    ("decode", "ffn_down"): (0, 34340864, "hbm"), ("decode", "qkv_proj"): (0, 13107200, "hbm"),
    ("decode", "o_proj"): (0, 8781824, "hbm"), ("decode", "lm_head"): (0, 541884416, "hbm"),
}
phase = None
print("| site | kernel | avg µs | work per launch | achieved | of peak |\n|---|---|---|---|---|---|")
for line in open(sys.argv[1]):
    m = re.match(r"== (\w+):", line)
    if m:
        phase = m.group(1); continue
    m = re.match(r"\s*[\d.]+ us/pass\s+n/pass=\s*[\d.]+\s+avg=\s*([\d.]+) us\s+(\w+)\s+(\S+)", line)
    if not m or (phase, m.group(2)) not in work:
        continue
    us, site, kern = float(m.group(1)), m.group(2), m.group(3).rstrip(",")
    fl, by, bound = work[(phase, site)]
    if bound == "mfma":
        tf = fl / us / 1e6
        print(f"| {phase} {site} | `{kern.split('<')[0]}` | {us:.1f} | {fl / 1e12:.4f} TFLOP | {tf:.0f} TF/s | {tf / 2500:.2f} |")
    else:
        tb = by / us / 1e6
        print(f"| {phase} {site} | `{kern.split('<')[0]}` | {us:.1f} | {by / 1e6:.1f} MB | {tb:.2f} TB/s | {tb / 8:.2f} |")
