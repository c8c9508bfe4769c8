"""Instruction classes of the LARGEST loop of a kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):
usage: python tools/isa_loop_classes.py file.s <substring of the kernel's mangled name> [dump]
Counts vector / matrix / LDS / memory / scalar / wait instructions of the loop body (static count = what one trip issues
when the loop has no inner branches) and lists the vector opcodes."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = len(sys.argv) > 3
    text = open(path).read()
    body = None
    for m in re.finditer(r"^(\S+):\s*;\s*@\S+\n(.*?)s_endpgm", text, re.S | re.M):
        if key in m.group(1):
            body = m.group(2)
            print("kernel", m.group(1)[:110])
            break
    if body is None:
        raise SystemExit("no kernel matches " + key)
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";") and (not l.strip().startswith(".") or l.strip().startswith(".LBB"))]
    lines = [l.split(";")[0].strip() for l in lines]
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r"s_cbranch_\w+ (\S+)", l) or re.match(r"s_branch (\S+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    loop = [l for l in lines[best[0]:best[1] + 1] if not l.endswith(":")]
    print("largest loop:", len(loop), "instructions")
    cnt, vops = collections.Counter(), collections.Counter()
    for l in loop:
        op = l.split()[0]
        if op.startswith("v_mfma"):
            k = "matrix"
        elif op.startswith("v_"):
            k = "vector"
            vops[op] += 1
        elif op.startswith("ds_"):
            k = "lds " + op
        elif op.startswith(("global_", "scratch_", "buffer_", "flat_", "tbuffer_")):
            k = "mem " + op
        elif op.startswith("s_waitcnt"):
            k = "s_waitcnt"
        elif op.startswith("s_nop"):
            k = "s_nop"
        elif op.startswith("s_barrier"):
            k = "s_barrier"
        else:
            k = "scalar"
        cnt[k] += 1
    for k, v in sorted(cnt.items(), key=lambda x: -x[1]):
        print(f"{v:6d}  {k}")
    print("vector opcodes:", ", ".join(f"{o} {n}" for o, n in vops.most_common()))
    if dump:
        print("\n".join(loop))


if __name__ == "__main__":
    main()
