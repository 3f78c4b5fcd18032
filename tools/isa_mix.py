#!/usr/bin/env python
"""Static instruction mix of k_sweep_kwt by section of the pass and by kind of instruction (round 6).

  python tools/isa_mix.py [out_dir | kwt_gfx950.elf] [kernel-name-substring] [--md] [--lines]

compiles kernels_kwt.hip as the Makefile does (plus -g), disassembles k_sweep_kwt (llvm-objdump) and asks llvm-symbolizer for the
inline stack of every instruction.  The kernel is one function with kwt_reach inlined three times (8-lane class B, 4-lane class C,
16-lane class A).  Every instruction is attributed
  * to a COPY by the template arguments of the kwt_reach frame of its inline stack (G = 4 / 8 / 16),
  * to a SECTION of the pass by the line of that frame (anchors looked up in kernels_kwt.hip, so the table follows edits): what an
    inlined helper (DPP reductions, ldx / stx, pow_0p4, kwt_wait_deps) costs is booked where it is called,
  * to a KIND by its mnemonic.
"Static" = every instruction counted once: one execution per pass of the straight-line path.  Loops (merge, thinning, interp sum) are
reported apart so that they can be weighted by trip counts (tools/kwt_sections.py / the particle counters give them).
"""
import collections
import os
import re
import sys

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mizuroute_amd", "csrc", "kernels_kwt.hip")

ANCHORS = [      # (section, text that starts it) in source order inside kwt_reach
    ("record", "---- round trip 1: the static record"),
    ("decode", "mzr_word wword = 0;"),
    ("wait", "// step t of this reach needs step t of every upstream reach"),
    ("loads", "---- round trip 2: everything that depends on the step"),
    ("need", "---- uniform: the work-array need"),
    ("stage", "// work arrays: a fixed slice of the wavefront's LDS pool"),
    ("merge", "// ---- qexmul_rch"),
    ("cold+neg", "if (cold) {   // getusq_rch"),
    ("thin", "// ---- remove_rch :999-1123"),
    ("wm", "// ---- extract_from_rch"),
    ("kinwav", "// ---- kinwav_rch :1130-1439 on particles"),
    ("count+qend", "// ---- time-step average and housekeeping"),
    ("interp", "const int _ibad = grp_interp_step"),
    ("results", "const double Qout = QNEW * rc[2] + ctx[1];"),
    ("outbox", "// record for the downstream reach"),
    ("atrest", "// at-rest state: KWAVE(NR+1:NQ2+1)"),
    ("publish", "if (PERS) {   // results written through"),
    ("END", "// Lane classes.  A routed reach is worked on"),
]
HELPER_SECTIONS = [      # helper functions whose own lines decide the section
    ("interp", "template <int G, int KS>\n__device__ __forceinline__ int grp_interp_step", "#ifdef MZR_KWT_TIMING"),
    ("interp", "template <int G, int KS>\n__device__ __forceinline__ int grp_interp_rch", "// interp_rch as kwt_rch calls it"),
    ("wait", "__device__ __forceinline__ bool kwt_wait_deps", "}  // namespace"),
    ("merge-serial", "__device__ __forceinline__ int kwt_merge_binary_serial", "// Confluences of more than two reaches are rare"),
    ("light", "template <bool FULL, bool PERS>\n__device__ __forceinline__ bool kwt_light", "}  // namespace"),
]


def source_sections():
    text = open(SRC).read()
    lines = text.splitlines()

    def line_of(snippet, start=0):
        pos = text.index(snippet, start)
        return text.count("\n", 0, pos) + 1

    body = []
    for name, snip in ANCHORS:
        body.append((line_of(snip), name))
    helpers = []
    for name, a, b in HELPER_SECTIONS:
        la = line_of(a)
        pos = text.index(a)
        lb = text.count("\n", 0, text.index(b, pos)) + 1
        helpers.append((la, lb, name))
    sweep0 = line_of("k_sweep_kwt(MzrDev dArg")
    return body, helpers, sweep0, len(lines)


def kind_of(m, ops):
    """(unit, kind) of an instruction"""
    if m.startswith("v_"):
        dpp = "dpp" in ops or "quad_perm" in ops or "row_" in ops
        if m.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "VALU", "readlane"
        if dpp:
            return "VALU", "dpp"
        if m.startswith("v_cmp") or m.startswith("v_cmpx"):
            return "VALU", "cmp_f64" if "f64" in m else "cmp_int"
        if m.startswith("v_cndmask"):
            return "VALU", "cndmask"
        if m.startswith(("v_mov", "v_accvgpr", "v_swap")):
            return "VALU", "mov"
        if "f64" in m:
            if m.startswith(("v_div_", "v_rcp", "v_rsq", "v_sqrt", "v_trig", "v_frexp", "v_ldexp")):
                return "VALU", "f64_div/rcp"
            return "VALU", "f64_arith"
        if "f32" in m or "f16" in m:
            return "VALU", "f32"
        if m.startswith(("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_mad_u64", "v_mad_i64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_lshl_add_u64", "v_add_u64")):
            return "VALU", "int64/addr"
        if m.startswith(("v_mbcnt", "v_bcnt", "v_ffb", "v_bfe", "v_bfi", "v_bfm", "v_alignbit", "v_perm")):
            return "VALU", "bit"
        if m.startswith(("v_cvt",)):
            return "VALU", "cvt"
        return "VALU", "int32"
    if m.startswith("s_"):
        if m.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_barrier", "s_endpgm", "s_sethalt", "s_setreg", "s_getreg", "s_memtime", "s_memrealtime", "s_code_end")):
            return "OTHER", m.split("_", 2)[1] if m.startswith("s_waitcnt") else "misc"
        if m.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
            return "SMEM", "s_load"
        if m.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call", "s_getpc")):
            return "SALU", "branch"
        if "saveexec" in m or "exec" in ops:
            return "SALU", "exec"
        if m.startswith("s_cmp") or m.startswith("s_bitcmp"):
            return "SALU", "s_cmp"
        if m.startswith(("s_mov", "s_cmov", "s_cselect")):
            return "SALU", "s_mov/sel"
        return "SALU", "s_alu"
    if m.startswith("ds_"):
        return "LDS", "ds_read" if ("read" in m or "load" in m) else "ds_write" if ("write" in m or "store" in m) else "ds_other"
    if m.startswith(("buffer_", "global_", "flat_", "scratch_")):
        if m.startswith("scratch_"):
            return "VMEM", "scratch"
        return "VMEM", "vm_load" if "load" in m else "vm_store" if "store" in m else "vm_atomic"
    return "OTHER", "misc"


LLVM = "/opt/rocm/lib/llvm/bin"


def build_elf(out_dir):
    """the kernel file compiled exactly as the Makefile compiles it, plus -g (line tables and inlined-call records; same code), unbundled"""
    import subprocess
    os.makedirs(out_dir, exist_ok=True)
    csrc = os.path.dirname(SRC)
    obj, elf = os.path.join(out_dir, "kwt_dev.o"), os.path.join(out_dir, "kwt_gfx950.elf")
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -disable-machine-licm".split()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + os.environ.get("EXTRA", "").split() + ["-g", "-c", "--offload-device-only", os.path.basename(SRC), "-o", obj],
                          cwd=csrc, stderr=subprocess.DEVNULL)
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
    return elf


def parse(elf, want):
    """instructions of the kernel whose mangled name contains `want`: (copy, section, unit, kind, mnemonic, line, addr, in_loop)"""
    import subprocess
    body, helpers, sweep0, nlines = source_sections()
    body_lines = [b[0] for b in body]
    rec_line, end_line = body[0][0], body[-1][0]
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", elf], capture_output=True, text=True).stdout.splitlines()
    name, rows, in_func = None, [], False
    for ln in dis:
        if not in_func:
            mm = re.match(r"^([0-9a-f]+) <(\S+)>:$", ln)
            if mm and want in mm.group(2) and mm.group(2).startswith("_Z"):
                in_func, name = True, mm.group(2)
            continue
        if re.match(r"^[0-9a-f]+ <", ln):
            break
        mm = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if mm:
            rows.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
    if not rows:
        raise SystemExit("kernel %s not found in %s" % (want, elf))
    # inline stacks of every instruction
    sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + elf, "--inlines", "-C"], input="\n".join(hex(a) for a, _, _ in rows) + "\n",
                         capture_output=True, text=True).stdout.split("\n\n")
    # inner loops: ranges of backward branches shorter than 700 instructions
    index = {a: i for i, (a, _, _) in enumerate(rows)}
    loops = []
    for i, (a, m, ops) in enumerate(rows):
        if m.startswith(("s_cbranch", "s_branch")):
            # objdump prints the branch's signed 16-bit offset in dwords, relative to the next instruction
            mm = re.match(r"\s*(\d+)\s*$", ops)
            tgt = None
            if mm:
                off = int(mm.group(1))
                if off >= 32768:
                    off -= 65536
                tgt = a + 4 + 4 * off
            if tgt is not None and tgt in index and index[tgt] <= i and i - index[tgt] < 700:
                loops.append((index[tgt], i))
    insts = []
    for i, (a, m, ops) in enumerate(rows):
        frames = []
        blk = sym[i].strip().splitlines() if i < len(sym) else []
        for k in range(0, len(blk) - 1, 2):
            fn, loc = blk[k], blk[k + 1]
            mm = re.match(r"(.*):(\d+):(\d+)$", loc)
            frames.append((fn, os.path.basename(mm.group(1)) if mm else "?", int(mm.group(2)) if mm else 0))
        copy, section, line = "-", "sweep", frames[-1][2] if frames else 0
        for fn, f, l in frames:      # innermost first: the kwt_reach frame gives the copy (G) and, by its line, the section
            if "kwt_reach<" in fn:
                G = fn.split("kwt_reach<")[1].split(",")[2].strip()
                copy = {"4": "C4", "8": "B8", "16": "A16"}.get(G, "G" + G)
                k = 0
                for j, bl in enumerate(body_lines):
                    if bl <= l:
                        k = j
                section, line = body[k][1], l
                break
        else:
            for fn, f, l in frames:
                if "kwt_light" in fn:
                    section = "light"
        for fn, f, l in frames[:1]:      # helpers with a section of their own (innermost frame)
            if f == "kernels_kwt.hip":
                hit = [h for h in helpers if h[0] <= l < h[1]]
                if hit and not (copy != "-" and hit[0][2] == "wait" and section != "wait"):
                    section = hit[0][2]
        unit, kind = kind_of(m, ops)
        inloop = [lp for lp in loops if lp[0] <= i <= lp[1]]
        insts.append((copy, section, unit, kind, m, line, a, min(inloop, key=lambda lp: lp[1] - lp[0]) if inloop else None))
    return name, insts


def table(rows, cols, get, title, md):
    out = [title]
    w = max(len(r) for r in rows) + 1
    if md:
        out.append("| " + " | ".join([""] + cols) + " |")
        out.append("|" + "---|" * (len(cols) + 1))
        for r in rows:
            out.append("| " + " | ".join([r] + [str(get(r, c)) for c in cols]) + " |")
    else:
        out.append(" " * w + "".join("%12s" % c for c in cols))
        for r in rows:
            out.append(r.ljust(w) + "".join("%12s" % get(r, c) for c in cols))
    return "\n".join(out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    md = "--md" in sys.argv
    elf = args[0] if args and os.path.exists(args[0]) else build_elf(args[0] if args else "/tmp/isa_mix")
    want = args[1] if len(args) > 1 else "k_sweep_kwtILb0ELi240E"
    name, insts = parse(elf, want)
    print("kernel", name, "instructions", len(insts))
    units = ["VALU", "SALU", "SMEM", "LDS", "VMEM", "OTHER"]
    copies = ["C4", "B8", "A16", "-"]
    by = collections.Counter()
    byl = collections.Counter()
    for copy, sec, unit, kind, m, line, a, lp in insts:
        by[(copy, sec, unit)] += 1
        if lp is not None:
            byl[(copy, sec, unit)] += 1
    order = [a[1] for a in ANCHORS[:-1]] + ["merge-serial", "sweep", "light"]
    secs = [x for x in order if any(i[1] == x for i in insts)] + sorted({i[1] for i in insts} - set(order))
    for copy in copies:
        rows = [x for x in secs if any(by[(copy, x, u)] for u in units)]
        if not rows:
            continue

        def get(r, c, copy=copy, rows=rows):
            tot = sum(by[(copy, x, c)] for x in rows) if r == "TOTAL" else by[(copy, r, c)]
            lo = sum(byl[(copy, x, c)] for x in rows) if r == "TOTAL" else byl[(copy, r, c)]
            return ("%d" % tot + (" (%d)" % lo if lo else "")) if tot else ""
        print()
        print(table(rows + ["TOTAL"], units, get, "## copy %s: static instructions by section and unit (in parentheses: of them inside an inner loop)" % copy, md))
    kinds = collections.Counter()
    for copy, sec, unit, kind, m, line, a, lp in insts:
        if unit in ("VALU", "SALU"):
            kinds[(copy, unit + " " + kind)] += 1
    klist = sorted({k for (_, k) in kinds}, key=lambda k: (k.split()[0] != "VALU", -sum(kinds[(c, k)] for c in copies)))
    print()
    print(table(klist, copies, lambda r, c: kinds[(c, r)] or "", "## VALU and SALU by kind (static)", md))
    vk = [k for k in klist if k.startswith("VALU")]
    for copy in ("A16", "B8", "C4"):
        ks = collections.Counter()
        for c2, sec, unit, kind, m, line, a, lp in insts:
            if unit == "VALU" and c2 == copy:
                ks[(sec, "VALU " + kind)] += 1
        rows = [x for x in secs if any(ks[(x, k)] for k in vk)]
        if rows:
            print()
            print(table(rows, [k.split()[1] for k in vk], lambda r, c: ks[(r, "VALU " + c)] or "", "## %s: VALU kind by section (static)" % copy, md))
    top = collections.Counter(m for copy, sec, unit, kind, m, line, a, lp in insts if unit in ("VALU", "SALU"))
    print()
    print("## most frequent VALU / SALU mnemonics (static, all copies):", ", ".join("%s %d" % kv for kv in top.most_common(40)))
    if "--lines" in sys.argv:      # hottest source lines of one copy
        cp = "A16"
        bl = collections.Counter((line, sec) for copy, sec, unit, kind, m, line, a, lp in insts if copy == cp and unit == "VALU")
        print()
        print("## A16: VALU instructions by source line of kwt_reach (top 60)")
        for (line, sec), n in bl.most_common(60):
            print("%5d %-12s %d" % (line, sec, n))


if __name__ == "__main__":
    main()
