"""A/B of kernel variants on the bench workload. Each variant is the in-tree library with ONE translation unit recompiled with extra
-D flags (debug switches that exist only under those macros), linked into tools/_ab/<name>/libmiphy.so; `run` then executes
`bench.py --no-cpu --no-extra --no-latency` once per variant (child processes, MIPHY_LIBRARY) and prints the per-kernel times.
  python tools/ab_bench.py build name=file.hip:-DA=1,-DB ...      (here or on the GPU box; hipcc cross-compiles)
  python tools/ab_bench.py run [name ...] [-- extra bench flags]
tools/_ab/ is git-ignored but travels to the GPU box."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "srsran_project_23.5_amd")
AB = os.path.join(ROOT, "tools", "_ab")


def build(specs):
    subprocess.check_call(["make", "-s", "-j8", "-C", PKG])
    for spec in specs:
        name, rest = spec.split("=", 1)
        src, _, flags = rest.partition(":")
        d = os.path.join(AB, name)
        os.makedirs(d, exist_ok=True)
        obj = os.path.join(d, src.replace(".hip", ".o"))
        per_file = ["-mllvm", "-enable-post-misched=false"] if src == "ldpc_decode_pk.hip" else []  # as the Makefile builds that file
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")]
                              + per_file + [f for f in flags.split(",") if f] + ["-c", os.path.join(PKG, "csrc", src), "-o", obj])
        others = [o for o in sorted(glob.glob(os.path.join(PKG, "build", "*.o"))) if os.path.basename(o) != os.path.basename(obj)]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libmiphy.so"), obj] + others)
        print("built", name)


def run(names, extra):
    names = names or sorted(os.listdir(AB))
    for name in ["(in-tree)"] + names:
        env = dict(os.environ)
        if name != "(in-tree)":
            env["MIPHY_LIBRARY"] = os.path.join(AB, name, "libmiphy.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extra", "--no-latency"] + extra, env=env, capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            print("%-18s %7.2f Gbit/s  %6.3f ms/step  %s  parity: %s" % (name, j["value"] / 1e9, j["ms_per_step"], {k: round(v, 4) for k, v in j["kernel_ms"].items()},
                                                                        str(j.get("parity_check"))[-5:]), flush=True)
        except Exception:
            print(name, "FAILED", r.stdout[-500:], r.stderr[-1500:], flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        a = sys.argv[2:]
        extra = a[a.index("--") + 1:] if "--" in a else []
        run(a[:a.index("--")] if "--" in a else a, extra)
