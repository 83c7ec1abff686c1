"""tests/golden/bus_480x640.jpg: a byte copy of the reference's demo image Assets/bus.jpg (480 x 640, w x h) -- the image
BASELINE config 1 names.  A data fixture (pixels, not source); copied here because /root/reference does not exist on the GPU box.
Network input shapes it yields: 640x480 (already multiples of 32; A = 6300) and, on a 114-padded square canvas, 640x640 (A = 8400).
    python tests/golden/make_bus_fixture.py"""
import os
import shutil
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
dst = os.path.join(HERE, "bus_480x640.jpg")
shutil.copyfile("/root/reference/Assets/bus.jpg", dst)
print("wrote", dst, os.path.getsize(dst), Image.open(dst).size)
