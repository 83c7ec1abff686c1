"""Print conv_p2_kernel's plan (tile, LDS bytes, pitches, modelled ds_read_b128 cycles) for the forward convolutions of YOLOv8n
B=64 640x640 -- host-only, uses the emulator build of the library (no device).  YS_P2_PITCH=0 YS_P2_ROWPAD=0 YS_P2_WPITCH=0 = round-2 rules."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "hipemu", "libyolosharp_emu.so"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = [  # (H, W, cin, cout, k, s, in_ldc)
    (640, 640, 8, 16, 3, 2, 8), (320, 320, 16, 32, 3, 2, 16), (160, 160, 32, 32, 1, 1, 32), (160, 160, 16, 16, 3, 1, 48), (160, 160, 48, 32, 1, 1, 48),
    (160, 160, 32, 64, 3, 2, 32), (80, 80, 64, 64, 1, 1, 64), (80, 80, 32, 32, 3, 1, 128), (80, 80, 128, 64, 1, 1, 128), (80, 80, 64, 128, 3, 2, 192),
    (40, 40, 128, 128, 1, 1, 128), (40, 40, 64, 64, 3, 1, 256), (40, 40, 256, 128, 1, 1, 256), (20, 20, 256, 256, 1, 1, 256), (20, 20, 128, 128, 3, 1, 384),
    (20, 20, 384, 256, 1, 1, 384), (20, 20, 512, 256, 1, 1, 512), (40, 40, 384, 128, 1, 1, 384), (40, 40, 192, 128, 1, 1, 192), (80, 80, 192, 64, 1, 1, 192),
    (80, 80, 96, 64, 1, 1, 96), (80, 80, 64, 64, 3, 2, 64), (80, 80, 64, 64, 3, 1, 64), (80, 80, 64, 80, 3, 1, 64), (80, 80, 80, 80, 3, 1, 80),
    (40, 40, 128, 64, 3, 1, 128), (40, 40, 128, 80, 3, 1, 128), (40, 40, 80, 80, 3, 1, 80), (20, 20, 80, 80, 3, 1, 80), (20, 20, 64, 64, 3, 1, 64),
    (80, 80, 64, 64, 1, 1, 64), (80, 80, 80, 80, 1, 1, 80), (40, 40, 128, 128, 3, 2, 128)]
buf = ctypes.create_string_buffer(512)
for (H, W, ci, co, k, s, ld) in shapes:
    lib.ys_debug_p2_plan(B, H, W, ci, co, k, s, ld, buf, 512)
    print(f"{H:3d}x{W:<3d} k{k} s{s} {ci:3d}->{co:<3d} : {buf.value.decode()}")
