"""Merge tile-table files (15 ints per line: 14-int shape key + choice); later files override earlier ones.

  python scripts/merge_tune.py cycle-diffusion_amd/tune_gfx950.txt gpurun_out/.../tune_new.txt [-o out.txt]

Used to fold the choices a GPU run appended to CYCLEDIFF_TUNE_CACHE into the shipped table.
"""
import sys


def main():
    args = sys.argv[1:]
    out = None
    if "-o" in args:
        i = args.index("-o")
        out = args[i + 1]
        args = args[:i] + args[i + 2:]
    table, order = {}, []
    for path in args:
        for ln in open(path):
            v = ln.split()
            if len(v) != 15:
                continue
            k = tuple(int(x) for x in v[:14])
            if k not in table:
                order.append(k)
            table[k] = int(v[14])
    lines = ["%s %d\n" % (" ".join(str(x) for x in k), table[k]) for k in order]
    with open(out or args[0], "w") as fh:
        fh.writelines(lines)
    print("%d entries -> %s" % (len(lines), out or args[0]))


if __name__ == "__main__":
    main()
