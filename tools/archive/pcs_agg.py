"""Aggregate rocprofv3 PC-sampling CSVs: samples by source line (Instruction_Comment), by instruction, by stall reason."""
import csv, glob, sys, collections, re
csv.field_size_limit(1 << 30)
root = sys.argv[1]
for f in glob.glob(root + "/**/*pc_sampling*.csv", recursive=True):
    rd = csv.DictReader(open(f))
    print("==", f, rd.fieldnames)
    byLine = collections.Counter(); byInst = collections.Counter(); byStall = collections.Counter(); byType = collections.Counter()
    issued = collections.Counter(); n = 0; byLineStall = collections.defaultdict(collections.Counter)
    for r in rd:
        n += 1
        c = r.get("Instruction_Comment", ""); ins = r.get("Instruction", "")
        m = re.search(r"([A-Za-z0-9_]+\.(?:hip|h)):(\d+)", c)
        key = (m.group(1) + ":" + m.group(2)) if m else c[-60:]
        byLine[key] += 1
        byInst[(key, ins)] += 1
        op = ins.split()[0] if ins else ""
        byType[op] += 1
        st = r.get("Stall_Reason") or r.get("Stall_Reason_Not_Issued") or ""
        if st: byStall[st] += 1; byLineStall[key][st] += 1
        w = r.get("Wave_Issued_Instruction") or r.get("Wave_Issued") or ""
        if w: issued[w] += 1
    print("samples", n)
    print("-- issued:", dict(issued))
    print("-- stall reasons:"); [print("  %8d %5.1f%% %s" % (v, 100.0 * v / max(n, 1), k)) for k, v in byStall.most_common(20)]
    print("-- opcodes:"); [print("  %8d %5.1f%% %s" % (v, 100.0 * v / max(n, 1), k)) for k, v in byType.most_common(40)]
    print("-- source lines:"); [print("  %8d %5.1f%% %s  %s" % (v, 100.0 * v / max(n, 1), k, dict(byLineStall[k].most_common(3)))) for k, v in byLine.most_common(150)]
    print("-- instructions:"); [print("  %8d %5.1f%% %s | %s" % (v, 100.0 * v / max(n, 1), k[0], k[1])) for k, v in byInst.most_common(150)]
