"""Static VALU opcode histogram of the DP kernels' loops, from the compiler's own assembly (make -C multiz_amd/csrc asm ->
mz_device.s): what bench.py prices `roofline.valu` with.  tests/tools/ub/ops.hip (profiles/r2_ub_ops.txt) measured two issue
rates on gfx950 at >= 2 waves per SIMD: v_add_u32 / v_sub_u32 / v_and_b32 at 2.3-2.5 shader cycles per wave-instruction, every
other integer VALU operation of this path (max, max3, and_or, alignbit, dot2/dot4, DPP forms, perm, bfe, mad24, compares) at
4.0-4.2.  For every kernel: all blocks LLVM marks as inside a loop, and the innermost loop with the most VALU instructions (the
steady-state row / step loop), each with its count of "fast" and other VALU instructions.

    python tests/tools/isa_hist.py [mz_device.s] > profiles/<tag>_isa_hist.json        (CPU only; run `make asm` first)
"""
import collections, hashlib, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FAST = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|and_b32|add_co_u32|sub_co_u32)(_e32|_e64)?$")      # (not the _dpp / _sdwa forms: ADDDPP measured 4.0)
CYC_FAST, CYC_REST = 2.35, 4.1


def sources_hash():
    h = hashlib.sha256()
    base = os.path.join(ROOT, "multiz_amd", "csrc")
    for f in [os.path.join(base, "mz_device.hip")] + sorted(glob.glob(os.path.join(base, "kernels", "*.inc"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "multiz_amd", "csrc", "mz_device.s")
    out = {"sources_hash": sources_hash(), "cycles": {"fast": CYC_FAST, "rest": CYC_REST},
           "fast_opcodes": "v_add_u32 v_sub_u32 v_subrev_u32 v_and_b32 v_add_co_u32 v_sub_co_u32 (e32 / e64 forms; DPP and SDWA forms priced with the rest)", "kernels": {}}
    name, loop, blocks = None, None, None
    for line in open(path):
        m = re.match(r"^(_Z\d+(k_\w+?)(?:I\w+E\w*)?\d*mz_\w+|_Z\d+(k_\w+)\w*):", line)
        if m and "@" in line:
            sym = re.search(r"@(\S+)", line).group(1)
            kn = re.match(r"_Z\d+(k_[a-z0-9_]+?)(?=\d\dmz_|I[a-z]E|P[Kx])", sym)
            name = kn.group(1) if kn else sym
            if "k_pre_ldsIs" in sym: name = "k_pre_lds<short>"
            if "k_pre_ldsIi" in sym: name = "k_pre_lds<int>"
            blocks = collections.defaultdict(lambda: [0, 0, collections.Counter()])     # loop header (or None) -> [valu, fast, opcodes]
            loop = None
            out["kernels"][name] = blocks
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        lb = re.match(r"^\.LBB\d+_(\d+):\s*(;.*)?$", line)
        if lb:
            c = lb.group(2) or ""
            if "This Inner Loop Header" in c or "This Loop Header" in c:
                loop = ("inner:" if "Inner" in c else "outer:") + lb.group(1)
            else:
                h = re.search(r"in Loop: Header=BB\d+_(\d+)", c) or re.search(r"Parent Loop", c) and None
                if h:
                    loop = next((k for k in blocks if k and k.endswith(":" + h.group(1))), "outer:" + h.group(1))
                elif "Loop" not in c:
                    loop = None
            continue
        op = line.strip().split()[0] if line.strip() and not line.strip().startswith((";", ".")) else None
        if op and op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            b = blocks[loop]
            b[0] += 1; b[1] += bool(FAST.match(op)); b[2][op] += 1
    res = {}
    for k, blocks in out["kernels"].items():
        inloop = [v for h, v in blocks.items() if h]
        inner = [(h, v) for h, v in blocks.items() if h and h.startswith("inner:")]
        if not inloop:
            continue
        tot = sum(v[0] for v in inloop); fast = sum(v[1] for v in inloop)
        hot = max(inner, key=lambda hv: hv[1][0]) if inner else None
        res[k] = {"in_loops": {"valu": tot, "fast": fast, "cycles_per_inst": round((fast * CYC_FAST + (tot - fast) * CYC_REST) / max(tot, 1), 4)}}
        if hot:
            v = hot[1]
            res[k]["largest_inner_loop"] = {"header": hot[0], "valu": v[0], "fast": v[1],
                                            "cycles_per_inst": round((v[1] * CYC_FAST + (v[0] - v[1]) * CYC_REST) / max(v[0], 1), 4),
                                            "opcodes": dict(v[2].most_common())}
    out["kernels"] = res
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
