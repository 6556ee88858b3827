"""CPU: the C-ABI shared library loads and exports every symbol include/edadm.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    from edadm import lib
    declared = lib.declared_symbols()
    assert len(declared) >= 40
    l = lib.load()                                  # raises if libedadm.so is missing
    for name in declared:
        assert hasattr(l, name), name
    assert l.edadm_abi_version() == 1
    assert l.edadm_reduce_ws_floats() > 0
    assert l.edadm_gn_ws_floats(2, 64, 64) == 2 * 8 * 64 * 2


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "edadm.h")).read()
    assert len(re.findall(r"\.py:\d+", src)) >= 15     # every entry point cites the reference interface it replaces


def test_bad_arguments_return_errno_not_crash():
    from edadm import lib
    l = lib.load()
    assert l.edadm_fake_quant_fwd(None, None, None, 10, None, None, 1, 1, 255.0, None, 1.0, 0, None) == -22
    assert l.edadm_qgemm_i8(None, 0, None, 0, 0, 0, 0, None, None, None, None, 0, None, 0, None, 0, None) == -22


def test_direct_conv_shape_gate():
    """edadm_conv3_direct_ok (host only): the shapes the direct 3x3 kernel takes -- 64-channel chunks, 192-column blocks (LDM-4 /
    LDM-8) or 128-column ones (the DDPM UNet, Stable Diffusion's 640 / 1280-channel levels; its 320-channel level takes the
    128-column tile with a zero-padded last block), widths 8 .. 64, power-of-two heights (its tile arithmetic is shifts), whole tiles of 256 or 128 pixels."""
    from edadm import lib
    ok = lib.load().edadm_conv3_direct_ok
    for B, H, W, Cin, N, want in ((100, 64, 64, 192, 192, 1), (100, 8, 8, 960, 960, 1), (4, 16, 16, 576, 192, 1),
                                  (3, 8, 8, 960, 960, 0),       # 3 images of 64 pixels fill neither 256- nor 128-pixel tiles
                                  (6, 8, 8, 960, 960, 1),       # ... 6 fill 128-pixel ones
                                  (2, 96, 64, 64, 192, 0),      # height no power of two
                                  (2, 64, 48, 64, 192, 0), (2, 64, 64, 96, 192, 0), (2, 64, 64, 64, 128, 1),
                                  (8, 32, 32, 640, 640, 1), (8, 16, 16, 2560, 1280, 1), (8, 64, 64, 320, 320, 1), (8, 64, 64, 320, 96, 0), (500, 32, 32, 128, 128, 1),
                                  (2, 64, 64, 64, 64, 0)):
        assert ok(B, H, W, Cin, N) == want, (B, H, W, Cin, N)
    tile = lib.load().edadm_conv3_direct_tile
    assert tile(100, 64, 64, 192, 192) == 256 and tile(100, 32, 32, 384, 384) == 256      # 1600 / 800 workgroups
    assert tile(100, 16, 16, 576, 576) == 256 and tile(100, 8, 8, 960, 960) == 128        # 300 / 125 at 256 pixels
    assert tile(6, 8, 8, 960, 960) == 128 and tile(3, 8, 8, 960, 960) == 0
