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
    assert lib.ec_version() == _lib.ABI_VERSION == header_abi_version()
    assert isinstance(lib.ec_last_error(), bytes)


def header_abi_version():
    text = open(os.path.join(ROOT, 'include', 'eventclip_hip.h')).read()
    return int(re.search(r'#define\s+EC_ABI_VERSION\s+(\d+)', text).group(1))


def test_abi_check_rejects_older_headers_and_short_structs():
    """ec_gemm_args / ec_vit_weights grew and ec_classify changed its argument list while ec_version() stayed
    100.  ABI 600: ec_abi_check compares the caller's header version and struct sizes with the library's (the loader
    of eventclip_amd/_lib.py calls it with its ctypes mirrors); a caller built against an older header is refused with a
    message, and the old ec_classify name is a stub that returns an error code."""
    lib = _lib.lib()
    sizes = [ctypes.sizeof(t) for t in (_lib.EcGemmArgs, _lib.EcBlockWeights, _lib.EcVitWeights, _lib.EcTextWeights,
                                        _lib.EcEventsParams, _lib.EcAdapterWeights)]
    assert lib.ec_abi_check(_lib.ABI_VERSION, *sizes) == 0
    assert lib.ec_abi_check(100, *sizes) == _lib.EC_ERR_INVALID and b'ABI 100' in lib.ec_last_error()
    short = list(sizes)
    short[0] = 176                                   # ec_gemm_args as round 4 had it (no A_lo / W_lo)
    assert lib.ec_abi_check(_lib.ABI_VERSION, *short) == _lib.EC_ERR_INVALID and b'ec_gemm_args' in lib.ec_last_error()
    short = list(sizes)
    short[2] -= 8                                    # ec_vit_weights without its last field
    assert lib.ec_abi_check(_lib.ABI_VERSION, *short) == _lib.EC_ERR_INVALID and b'ec_vit_weights' in lib.ec_last_error()
    assert lib.ec_classify() == _lib.EC_ERR_UNSUPPORTED and b'ec_classify_v2' in lib.ec_last_error()


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.EcFrameStats) == 40
    assert ctypes.sizeof(_lib.EcEventsParams) == 80
    assert ctypes.sizeof(_lib.EcGemmArgs) == 248     # (+ A_lo, W_lo: round 5; + the four e4m3 operands and their exponents: round 6)


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


def test_gemm_kernels_keep_their_schedule_and_have_no_scratch():
    """csrc/gemm.hip's persistent kernel issues the MFMAs of a phase between `s_setprio 1` / `s_setprio 0`, closed by barriers,
    and counts its vmcnt waits: tools/check_gemm_isa.py compiles the file and checks every instantiation for scratch, for an
    empty matrix-pipe slot (round 6: hipcc had sunk the e4m3 MFMAs of a phase in front of the next phase's) and for phases
    whose MFMA count is neither 32 (16-bit) nor 16 (e4m3)."""
    import shutil
    import sys
    import pytest
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip('no hipcc')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_gemm_isa
    assert check_gemm_isa.main() == 0
