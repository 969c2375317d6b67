"""Static check of the compiled GEMM kernels (gfx950 ISA): the schedule the persistent kernel is written for must survive hipcc.

    python tools/check_gemm_isa.py        # compiles csrc/gemm.hip with -save-temps into a temp dir

For every `gemm2pp_kernel<...>` instantiation of the product build it reports
  * scratch (private segment) use: a spilled value comes back behind `s_waitcnt vmcnt(0)`, which also waits for the staging
    DMA in flight -- every counted wait of the pipeline loses its meaning;
  * an EMPTY matrix-pipe slot -- `s_setprio 1` directly followed by `s_setprio 0`: the MFMAs of a phase are issued between
    those two, closed by barriers on both sides; round 6 found hipcc sinking the sixteen e4m3 MFMAs of a phase (a pure
    intrinsic whose results are read an iteration later) out of that slot in front of the next phase's, which cost the e4m3
    K tile 60 % (profiles/r6_gemm.md 1);
  * a K-loop phase whose MFMA count is neither 32 (16-bit: v_mfma_f32_16x16x32) nor 16 (e4m3: v_mfma_scale_f32_16x16x128).
Exit code 1 if anything was reported.  Run by tests/test_capi_symbols.py (CPU suite) when hipcc is present.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'eventclip_amd', 'csrc', 'gemm.hip')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def check(asm_text):
    problems, n = [], 0
    for m in re.finditer(r'^(_ZN12_GLOBAL__N_114gemm2pp_kernel\w+):', asm_text, re.M):
        name = m.group(1)
        end = asm_text.index('.end_amdhsa_kernel', m.start())
        text = asm_text[m.start():end]
        n += 1
        sc = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', text)
        if sc and int(sc.group(1)) != 0:
            problems.append(f'{name}: {sc.group(1)} bytes of scratch')
        body = [ln.strip() for ln in text.split('\n') if ln.strip() and not ln.strip().startswith(';')]
        for a, b in zip(body, body[1:]):
            if a == 's_setprio 1' and b == 's_setprio 0':
                problems.append(f'{name}: an empty matrix-pipe slot (s_setprio 1 directly followed by s_setprio 0)')
                break
        # MFMAs between a `s_setprio 1` and the next `s_setprio 0`
        count, inside = 0, False
        for ins in body:
            if ins == 's_setprio 1':
                inside, count = True, 0
            elif ins == 's_setprio 0' and inside:
                inside = False
                if count not in (16, 32):
                    problems.append(f'{name}: a phase with {count} MFMAs between s_setprio 1 and s_setprio 0')
                    break
            elif inside and ins.startswith('v_mfma'):
                count += 1
    if n < 20:
        problems.append(f'only {n} gemm2pp_kernel instantiations found: the label pattern no longer matches')
    return problems, n


def main():
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-fno-gpu-rdc',
               '-I', os.path.join(ROOT, 'include'), '-I', os.path.dirname(SRC), '-save-temps', '-c', SRC, '-o', os.path.join(tmp, 'gemm.o')]
        subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')]
        assert asm, os.listdir(tmp)
        problems, n = check(open(os.path.join(tmp, asm[0])).read())
    for p in problems:
        print(p)
    print(f'{n} gemm2pp_kernel instantiations checked:', 'OK' if not problems else f'{len(problems)} problem(s)')
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
