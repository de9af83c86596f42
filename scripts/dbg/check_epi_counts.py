#!/usr/bin/env python3
"""Build-time check of the counted s_waitcnt vmcnt(N) waits that let the GEMM tails' global loads fly under the last
operand stages (csrc/gemm_nt.hip, csrc/gemm_ln.hip, csrc/gemm_lnbwd.hip).

Those kernels wait for their last global_load_lds stage with `s_waitcnt vmcnt(EARLY)`, EARLY = the number of tail loads
issued behind that stage.  That is only correct if the compiler emitted EXACTLY that many vector-memory loads between
the preceding barrier and the wait (fewer => the stage may not have landed when it is read).  This script compiles the
two files to ISA and, in every kernel, checks for each `s_waitcnt vmcnt(N)`, N > 0, that follows register loads:
    #register loads (global_load_* without lds, buffer_load_*) since the previous s_barrier  ==  N
and that the main loops contain no vmcnt(0).  Run by tests/test_host_cpu.py (CPU, hipcc cross-compiles)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "vit-vs-raw-iq_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def isa(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
                           "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(line)
            if "s_endpgm" in line:
                yield cur, body
                cur = None


def check(src, want_prefix):
    """Within one basic block:  [s_barrier] L register loads ... s_waitcnt vmcnt(N) s_barrier.
    The wait must retire the last operand stage, which is OLDER than the L loads: safe iff N <= L, or the compiler put
    a vmcnt(0) between (then nothing older is left).  N < L or such a drain is only slower, and is reported."""
    problems, notes, seen = [], [], 0
    for name, body in kernels(isa(src)):
        if want_prefix not in name:
            continue
        loads, loads_bar, drained = 0, 0, False     # register loads since the block start / since the last s_barrier
        counted = []
        for idx, ln in enumerate(body):
            t = ln.strip()
            if t.startswith("s_barrier"):
                loads, loads_bar, drained = 0, 0, False
            elif re.match(r"\.LBB\d+_\d+:", t) or t.startswith(("s_cbranch", "s_branch")):
                loads = 0           # (a forward branch around a wave-dependent MFMA block may sit between the loads and the wait)
            elif t.startswith("global_load_lds"):
                loads, loads_bar, drained = 0, 0, False      # loads issued before a DMA are OLDER than it: they do not count
            elif re.match(r"(global_load_(dword|ubyte|ushort|short)|buffer_load_)", t):
                loads += 1
                loads_bar += 1
            else:
                m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    nxt = [x.strip() for x in body[idx + 1: idx + 4]]
                    if n == 0:
                        drained = loads_bar > 0 or drained
                    elif (loads_bar > 0 or drained) and any(x.startswith("s_barrier") for x in nxt):
                        have = max(loads, loads_bar)
                        counted.append((n, have, "drained" if drained else ""))
                        if not drained and n > have:
                            problems.append(f"{name}: s_waitcnt vmcnt({n}) + s_barrier behind only {have} register loads: "
                                            "the last operand stage may not have landed")
                        elif drained or n != have:
                            notes.append(f"{name}: tail wait vmcnt({n}) with {have} loads{' after a full drain' if drained else ''} (safe, not fully overlapped)")
        seen += 1
        m_epi = re.search(r"gemm_nt_async_kernelILi\d+ELi\d+ELi(\d+)ELb0", name)
        expects = "gemm_ln_kernel" in name or "gemm_lnbwd_kernel" in name or (m_epi and int(m_epi.group(1)) & (1 | 2))    # residual / gate tails
        if expects and not counted:
            problems.append(f"{name}: no counted tail wait found")
        print(f"{name[:90]:90s} counted tail waits {counted}")
    for n_ in notes:
        print("note:", n_)
    return seen, problems


def main():
    total, problems = 0, []
    for src, pref in (("gemm_nt.hip", "Lb0EEEv10GemmParams"), ("gemm_ln.hip", "gemm_ln_kernel"),      # Lb0 = the non-RESK instantiations
                      ("gemm_lnbwd.hip", "gemm_lnbwd_kernel")):
        n, pr = check(src, pref)
        total += n
        problems += pr
    if problems:
        print("\n".join(problems))
        return 1
    print(f"ok: {total} kernels, every counted tail wait matches the number of loads issued behind the last stage")
    return 0


if __name__ == "__main__":
    sys.exit(main())
