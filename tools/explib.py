"""Tools that flip experiment switches (brick layout / classes / debug flags, XCD mapping) or read
the phase profile run against a TOOLS build of the kernels, never the product library:
    import tools.explib; tools.explib.use("exp")      # -DDDRR_EXPERIMENTS
    import tools.explib; tools.explib.use("prof")     # -DDDRR_BRICK_PROFILE
builds tools/_build/libdiffdrr_hip_<name>.so when stale (hipcc) and points diffdrr_amd._lib at it.
Call before anything touches diffdrr_amd.ops."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
FLAGS = {"exp": ["-DDDRR_EXPERIMENTS"], "prof": ["-DDDRR_BRICK_PROFILE"]}


def use(name="exp", extra_flags=()):
    import __graft_entry__ as ge
    from diffdrr_amd import _lib

    # DDRR_EXP_FLAGS="-DDDRR_WALK_CHECK4 ...": one more build variant, named after the flags
    env = os.environ.get("DDRR_EXP_FLAGS", "").split()
    base = FLAGS.get(name, [])
    if env:
        extra_flags = tuple(extra_flags) + tuple(env)
        name = name + "_" + "_".join(f.replace("-D", "").lower() for f in env)

    out = os.path.join(ROOT, "tools", "_build", f"libdiffdrr_hip_{name}.so")
    srcs = [os.path.join(ge.CSRC, f) for f in ge.HIP_SOURCES]
    deps = srcs + [os.path.join(ge.CSRC, f) for f in ge.HIP_HEADERS]
    if not (os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps)):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run([ge._hipcc(), *ge.HIP_FLAGS, *base, *extra_flags, "-shared", *srcs,
                        "-o", out], check=True, cwd=ge.CSRC)
    _lib.LIB_PATH = os.environ.get("DDRR_LIB", out)
    return out


if __name__ == "__main__":
    for n in sys.argv[1:] or ["exp", "prof"]:
        print(use(n))
