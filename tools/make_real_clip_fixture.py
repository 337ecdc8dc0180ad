#!/usr/bin/env python
"""Writes tests/golden/real_v32/: a small, down-scaled excerpt of the reference's own sample inputs
(/root/reference/sample_videos/clips/v32/*.png: 1280x720 frames of a monochrome film scan, 37 % of their pixels at or
near black — flat regions, exact and near ties in the correlation, saturated shadows; /root/reference/sample_videos/ref/
v32/*.jpg: the four colour references test.py:169-181 loops over) for the real-content parity tests
(tests/test_gpu_real_clip.py).  /root/reference does not exist on the GPU box, so the excerpt is committed:

  frames  the first N_FRAMES frames in test.py:41's numeric order, luminance only (the scan's R/G/B differ by <= 7
          levels; the colourisation path reads L alone), Lanczos-resampled to 960x540 — still LARGER than the 768x432 the
          ingest produces, so CenterPad's anti-aliased down-scale (utils/util_distortion.py:217-258) is exercised —
          8-bit grayscale PNG (the tests replicate the plane to RGB);
  refs    the four references, thumbnailed to <= 360 px, RGB PNG.

These are reference INPUTS (sample data, not code); ~1.5 MB.  Run in the build container:  python tools/make_real_clip_fixture.py
"""
import os

from PIL import Image

SRC = "/root/reference/sample_videos"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real_v32")
N_FRAMES = 10


def main():
    os.makedirs(os.path.join(DST, "clip"), exist_ok=True)
    os.makedirs(os.path.join(DST, "ref"), exist_ok=True)
    clip = os.path.join(SRC, "clips", "v32")
    names = sorted(os.listdir(clip), key=lambda f: int("".join(filter(str.isdigit, f) or -1)))      # test.py:41
    total = 0
    for name in names[:N_FRAMES]:
        im = Image.open(os.path.join(clip, name)).convert("L").resize((960, 540), Image.LANCZOS)
        out = os.path.join(DST, "clip", name)
        im.save(out, format="PNG", optimize=True)
        total += os.path.getsize(out)
    for name in sorted(os.listdir(os.path.join(SRC, "ref", "v32"))):
        im = Image.open(os.path.join(SRC, "ref", "v32", name)).convert("RGB")
        im.thumbnail((360, 360), Image.LANCZOS)
        out = os.path.join(DST, "ref", os.path.splitext(name)[0] + ".png")
        im.save(out, format="PNG", optimize=True)
        total += os.path.getsize(out)
    print(f"wrote {DST}: {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
