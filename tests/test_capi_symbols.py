"""The C-ABI library loads on a CPU-only box and exports every symbol the header declares."""
import ctypes
import os
import re

from eventclip_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'eventclip_hip.h')).read()
    return sorted(set(re.findall(r'EC_API\s+[\w\s\*]+?\b(ec_\w+)\s*\(', text)))


def test_library_built_and_exports_header():
    syms = header_symbols()
    assert len(syms) >= 4
    assert os.path.exists(_lib.LIB_PATH), 'run `python -m eventclip_amd.build`'
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(handle, s), f'{s} declared in include/eventclip_hip.h but not exported'


def test_binding_table_covers_header():
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_version_and_error_string():
    lib = _lib.lib()
    assert lib.ec_version() >= 100
    assert isinstance(lib.ec_last_error(), bytes)


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.EcFrameStats) == 40
    assert ctypes.sizeof(_lib.EcEventsParams) == 80
    assert ctypes.sizeof(_lib.EcGemmArgs) == 192     # (+ A_lo, W_lo: round 5)


def test_attention_kernels_have_no_unpadded_asm_hazards():
    """csrc/attention.hip updates its accumulators with inline-asm MFMAs, whose hazards hipcc does not pad
    (cdna_hip_programming.md 5.7): tools/check_attn_isa.py compiles the file and scans every product kernel for a
    vector write right in front of an unpadded asm MFMA and for accumulator copies in the key-block loops."""
    import os
    import shutil
    import sys
    import pytest
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip('no hipcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import check_attn_isa
    assert check_attn_isa.main() == 0
