"""Static check of the compiled attention kernels (gfx950 ISA) for the hazards hipcc does not pad around the
in-place asm MFMAs of csrc/attention.hip (cdna_hip_programming.md 5.7 item 2).

    python tools/check_attn_isa.py        # compiles csrc/attention.hip with -save-temps into a temp dir

For every `attention_kernel<...>` it walks each basic block and reports
  * an asm MFMA (between ;;#ASMSTART / ;;#ASMEND) that has NO `s_nop` inside its string and whose operand
    registers were written by a vector-ALU instruction within the two instructions in front of it;
  * vector copies (v_mov_b32 / v_mov_b64 / v_accvgpr) inside the innermost key-block loops (the thing the asm
    accumulators exist to avoid): more than four per loop body is reported.
Exit code 1 if anything was reported.  Run by tests/test_capi_symbols.py (CPU suite) when hipcc is present.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'eventclip_amd', 'csrc', 'attention.hip')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def is_valu_write(ins):
    op = ins.split()[0]
    return op.startswith('v_') and not op.startswith(('v_mfma', 'v_cmp', 'v_nop', 'v_readfirstlane'))


def written(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return set()
    return regs(parts[1].split(',')[0])


def check(asm_text):
    problems = []
    kernels = re.split(r'\n(?=_ZN[^\n]*attention_kernel[^\n]*:\s*;)', asm_text)
    for k in kernels:
        head = k.split('\n', 1)[0]
        if 'attention_kernel' not in head or 'Lb1EEEvNS' not in head.replace(' ', ''):
            # only the product variant (V2 = true, last template argument)
            if 'attention_kernel' not in head:
                continue
        name = head.split(':')[0]
        if not name.endswith('Lb1EEEvNS_8AttnArgsE'):
            continue
        body = k.split('.Lfunc_end')[0].split('\n')
        prev = []          # last real instructions of the current block
        in_asm = False
        asm_lines = []
        loop_movs = {}
        cur_label, loops = None, {}
        for line in body:
            t = line.strip()
            if not t or t.startswith(';') and 'ASMSTART' not in t and 'ASMEND' not in t:
                continue
            if t.startswith('.LBB'):
                cur_label = t.split(':')[0]
                inner = 'Inner Loop Header' in t or ('Depth=2' in t and 'Parent Loop' in t)
                loops[cur_label] = inner
                prev = []
                continue
            if 'ASMSTART' in t:
                in_asm, asm_lines = True, []
                continue
            if 'ASMEND' in t:
                in_asm = False
                mf = [x for x in asm_lines if x.startswith('v_mfma')]
                if mf and not any(x.startswith('s_nop') for x in asm_lines):
                    need = set()
                    for x in mf:
                        need |= regs(x.split(None, 1)[1])
                    for p in prev[-2:]:
                        if is_valu_write(p) and written(p) & need:
                            problems.append(f'{name}: unpadded asm MFMA `{mf[0]}` right behind `{p}`')
                prev = prev + asm_lines
                continue
            if in_asm:
                asm_lines.append(t)
                continue
            ins = t.split(';')[0].strip()
            if not ins:
                continue
            if ins.split()[0] in ('v_mov_b32_e32', 'v_mov_b64_e32', 'v_mov_b32', 'v_mov_b64') or ins.startswith('v_accvgpr'):
                if cur_label and loops.get(cur_label):
                    loop_movs[cur_label] = loop_movs.get(cur_label, 0) + 1
            prev.append(ins)
        for lab, n in loop_movs.items():
            if n > 4:
                problems.append(f'{name}: {n} vector copies in loop block {lab}')
    return problems


def main():
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-fno-gpu-rdc',
               '-I', os.path.join(ROOT, 'include'), '-save-temps', '-c', SRC, '-o', os.path.join(tmp, 'a.o')]
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            return 2
        s_file = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')][0]
        problems = check(open(os.path.join(tmp, s_file)).read())
    for p in problems:
        print(p)
    print(f'{len(problems)} problem(s)')
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
