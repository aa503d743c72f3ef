"""Builds libdspeed_hip.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m dspeed_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.

    python -m dspeed_amd.build --diag

builds libdspeed_hip_diag.so beside it: the same sources with -DDSPEED_HIP_DIAG, which compiles the kernels' diagnostic switches
in (DSPEED_HIP_ABLATE: skip passes / per-phase cycle stamps).  The product library does not contain them; tools select the
diagnostic one with DSPEED_HIP_LIB=.../libdspeed_hip_diag.so.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdspeed_hip.so")
LIB_DIAG = os.path.join(HERE, "libdspeed_hip_diag.so")
SOURCES = ["dsp_vm.hip", "dsp_energy.hip", "dsp_fit.hip", "dsp_rows.hip", "dsp_current.hip", "dsp_scalar.hip", "dsp_reduce.hip", "dsp_pz.hip", "dsp_fir_mfma.hip", "dsp_fir_f16.hip", "dsp_fir_runs.hip", "dsp_plan.cpp", "dsp_host.cpp"]
DEPS = SOURCES + ["dsp_program.h", "dsp_plan.h", "dsp_wave.h", "dsp_reduce_tail.h", os.path.join("..", "..", "include", "dspeed_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function"]


#: kernels that must not use scratch memory: name fragment -> why.  The waveform VM keeps its interpreter state (Ctx) in registers; it
#: lands in scratch memory -- and every op of every chain pays -- as soon as a helper that takes the state by reference is not inlined
#: (the compiler decides by size: one more branch in op_pickoff did it once, C2 on the VM 129 -> 113 M waveforms/s) or an indexed local
#: array appears.  The build fails rather than ship that.
NO_SCRATCH = {"dsp_vm.hip": "dsp_vm_kernel"}


def _check_scratch(src: str, remarks: str) -> None:
    frag, name, bad = NO_SCRATCH[src], None, []
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize [bytes/lane]:" in line and name and frag in name:
            n = int(line.split("ScratchSize [bytes/lane]:")[1].split()[0])
            if n:
                bad.append((name, n))
    skip = 0
    for line in remarks.splitlines():  # (the compiler's other diagnostics stay visible; a remark is followed by its source excerpt)
        if "remark:" in line:
            skip = 2
        elif skip and (line.lstrip()[:1].isdigit() or line.lstrip().startswith("|")):
            skip -= 1
        else:
            skip = 0
            sys.stderr.write(line + "\n")
    if bad:
        raise RuntimeError(f"{src}: scratch memory in " + ", ".join(f"{n} ({b} B/lane)" for n, b in bad) +
                           " -- a helper taking Ctx& was outlined, or a local array is indexed at run time (see NO_SCRATCH in build.py)")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm)")


def _stamp(diag: bool) -> str:
    """what a library was built from: the compiler flags, the scratch guard and the content of every source and header"""
    import hashlib

    h = hashlib.sha256(repr((FLAGS, sorted(NO_SCRATCH.items()), diag)).encode())
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    return h.hexdigest()


def _stamp_file(lib: str) -> str:
    return lib + ".stamp"


def needs_build(lib: str = LIB) -> bool:
    """True unless `lib` exists and was built from exactly these sources with exactly these flags (a stamp file beside it says so)"""
    if not os.path.exists(lib) or not os.path.exists(_stamp_file(lib)):
        return True
    with open(_stamp_file(lib)) as f:
        return f.read().strip() != _stamp(lib == LIB_DIAG)


def build(force: bool = False, verbose: bool = False, diag: bool = False, variant: str | None = None, defines=(), extra_flags=()) -> str:
    """``variant`` / ``defines``: an experiment library libdspeed_hip_<variant>.so compiled with -D<define>... beside the product one
    (A/B runs of kernel variants in one GPU session: tools select it with DSPEED_HIP_LIB); always rebuilt."""
    lib = LIB_DIAG if diag else LIB
    if variant:
        lib = os.path.join(HERE, f"libdspeed_hip_{variant}.so")
        force = True
    if not force and not needs_build(lib):
        return lib
    # one hipcc process per source (they are independent translation units), then a link: the wall time of the longest file
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, "." + src + (f".{variant}.o" if variant else (".diag.o" if diag else ".o")))
        cmd = [_hipcc(), *[f for f in FLAGS if f != "-shared"], *(["-DDSPEED_HIP_DIAG"] if diag else []), *[f"-D{d}" for d in defines], *extra_flags, "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj]
        guard = src in NO_SCRATCH
        if guard:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE if guard else None, text=True if guard else None), src))
        objs.append(obj)
    failed = None
    try:
        for cmd, p, src in procs:
            err = p.communicate()[1] if src in NO_SCRATCH else None
            if p.wait() != 0:
                if err:
                    sys.stderr.write(err)
                raise subprocess.CalledProcessError(p.returncode, cmd)
            if err is not None:
                _check_scratch(src, err)
    except BaseException as e:
        failed = e
    if failed is not None:  # stop the compiles still running and leave no object behind for a later link to pick up
        for _cmd, p, _src in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        for obj in objs:
            if os.path.exists(obj):
                os.remove(obj)
        raise failed
    link = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", lib]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    if variant:
        for obj in objs:
            os.remove(obj)
        return lib
    with open(_stamp_file(lib), "w") as f:
        f.write(_stamp(diag) + "\n")
    return lib


if __name__ == "__main__":
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    _defines = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--define"]
    _flags = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--flag"]  # raw compiler flags of an experiment library
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv, variant=_variant, defines=_defines, extra_flags=_flags))
