#!/usr/bin/env python3
"""Developer aid: register / spill / LDS report of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage), one line per
kernel. usage: scripts/kres.py [file.hip ...] [--filter substr] [--defs "-DX=1 ..."]   (default: every csrc/*.hip)"""
import glob, os, re, subprocess, sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hybvio_amd", "csrc")


def demangle(names):
    try:
        out = subprocess.run(["/usr/bin/c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def report(path, defs=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", path, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *defs]
    txt = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    rows, cur = [], None
    for line in txt.splitlines():
        m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def main():
    args = sys.argv[1:]
    flt, defs, files = None, [], []
    while args:
        a = args.pop(0)
        if a == "--filter": flt = args.pop(0)
        elif a == "--defs": defs = args.pop(0).split()
        else: files.append(a)
    files = files or sorted(os.path.basename(f) for f in glob.glob(os.path.join(CSRC, "*.hip")))
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>3s} {'LDS':>7s}")
    for f in files:
        rows = report(f, defs)
        dm = demangle([r["name"] for r in rows])
        for r in rows:
            name = re.sub(r"hv::\(anonymous namespace\)::|\(anonymous namespace\)::|hv::", "", dm.get(r["name"], r["name"]))
            name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
            if flt and flt not in name:
                continue
            print(f"{(f[:-4] + ':' + name)[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} "
                  f"{r.get('SGPRs Spill', '?'):>6s} {r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>3s} {r.get('LDS Size [bytes/block]', '?'):>7s}")


if __name__ == "__main__":
    main()
