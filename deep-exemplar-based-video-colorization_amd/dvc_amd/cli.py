"""The clip driver of /root/reference/test.py on top of the MI355X hot path.

`colorize_video(opt, input_path, reference_file, output_path, nonlocal_net, colornet, vggnet)` has the reference's
signature (test.py:29) and does what test.py:29-124 does: walk the clip folder in numeric file order, take the reference
image (or the first frame with --frame_propagate), colourise frame after frame with the recurrence of test.py:76-96,
WLS-filter, convert to 8-bit RGB, write `00000.jpg ...` and `video.avi` into `output_path`.  What differs is where the
work happens: frames are decoded on the host (PIL), uploaded as 8-bit RGB, and everything between `Image.open` and the
JPEG encoder runs on the device in batches of `opt.batch_frames` frames (ClipColorizer.colorize_video: CenterPad,
RGB->Lab, x0.5, VGG19 + WarpNet + fused correlation + ColorVidNet with the exemplar side cached per clip and the next
frames' front ends on side HIP streams, x2 bilinear * 1.25, fast global smoother, Lab->RGB8).

`main()` keeps the reference's flags WITH their quirks (test.py:127-135):
  --frame_propagate  `type=bool`: ANY non-empty value enables it ("--frame_propagate False" is True), as upstream;
  --image_size       `type=int` with a two-element list default: passing a value on the command line yields a single int
                     and fails in CenterPad exactly as upstream — leave it at its default (432 x 768);
  --cuda (store_false), --gpu_ids: parsed, printed, otherwise unused — as upstream;
  the output folder is `<output_path>/<clip>_<reference stem>` and every reference image of --ref_path is tried, errors
  are printed and skipped (test.py:168-181).
Extra, optional flags (not upstream): --vgg_path/--nonlocal_path/--colornet_path (the upstream paths are the defaults),
--synthetic_weights (no checkpoints at hand: deterministic synthetic weights), --batch_frames, --refs_per_pass.

The per-reference loop (test.py:169-181: the SAME clip colourised once per reference image; the sample set ships 3-6
references per clip) is re-designed rather than transcribed: R references are R independent recurrences over the same
frames, so `colorize_video_refs` decodes / ingests every frame ONCE, runs ONE VGG19 + WarpNet front end per frame, R fused
correlations (theta shared) and the ColorVidNet chain at batch R (ClipColorizer.set_exemplars), and writes the R output
folders the loop would have written.  Opt-in: `--refs_per_pass R` (default 1 = one pass per reference, as upstream).
"""
import argparse
import glob
import io
import os
import struct

import numpy as np
import torch

_VIDEO_FPS = 24          # utils/util.py:262


def mkdir_if_not(dir_path):
    """utils/util.py:287-289"""
    if not os.path.exists(dir_path):
        os.makedirs(dir_path)


def save_frames(image, image_folder, index=None, image_name=None):
    """utils/util.py:246-252 (skimage.io.imsave -> PIL, same file names; JPEG at PIL's default quality 75, which is
    what skimage's default imageio/PIL plugin writes)."""
    from PIL import Image
    if image is not None:
        image = np.clip(image, 0, 255).astype(np.uint8)
        name = image_name if image_name else str(index).zfill(5) + ".jpg"
        Image.fromarray(image).save(os.path.join(image_folder, name))


def write_mjpeg_avi(path, jpeg_frames, width, height, fps=_VIDEO_FPS):
    """A minimal RIFF/AVI writer with one MJPG video stream (every frame an independent JPEG, 'idx1' index)."""
    frames = [bytes(f) for f in jpeg_frames]

    def chunk(tag, data):
        return tag + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b"")

    def lst(tag, data):
        return b"LIST" + struct.pack("<I", len(data) + 4) + tag + data

    n = len(frames)
    maxb = max((len(f) for f in frames), default=0)
    avih = struct.pack("<14I", int(1e6 / fps), maxb * fps, 0, 0x10, n, 0, 1, maxb, width, height, 0, 0, 0, 0)
    strh = (b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, fps, 0, n, maxb, 0xFFFFFFFF, 0) +
            struct.pack("<4h", 0, 0, width, height))
    strf = struct.pack("<IiiHH4sIiiII", 40, width, height, 1, 24, b"MJPG", width * height * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi_body, idx, off = b"", b"", 4
    for f in frames:
        c = chunk(b"00dc", f)
        idx += b"00dc" + struct.pack("<III", 0x10, off, len(f))
        movi_body += c
        off += len(c)
    body = b"AVI " + hdrl + lst(b"movi", movi_body) + chunk(b"idx1", idx)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def folder2vid(image_folder, output_dir, filename):
    """utils/util.py:255-268: all .jpg/.png of `image_folder`, sorted, into `<output_dir>/<filename>` at 24 fps.
    With OpenCV present this is the reference's own DIVX writer; without it (this image) an MJPG AVI is written."""
    from PIL import Image
    images = [img for img in os.listdir(image_folder) if img.endswith(".jpg") or img.endswith(".png")]
    images.sort()
    print("writing to video " + os.path.join(output_dir, filename))
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None and hasattr(cv2, "VideoWriter"):
        frame = cv2.imread(os.path.join(image_folder, images[0]))
        height, width, _ = frame.shape
        video = cv2.VideoWriter(os.path.join(output_dir, filename), cv2.VideoWriter_fourcc("D", "I", "V", "X"),
                                _VIDEO_FPS, (width, height))
        for image in images:
            video.write(cv2.imread(os.path.join(image_folder, image)))
        video.release()
        return
    jpegs, size = [], None
    for name in images:
        im = Image.open(os.path.join(image_folder, name)).convert("RGB")
        size = size or im.size
        if im.size != size:
            im = im.resize(size)
        buf = io.BytesIO()
        im.save(buf, format="JPEG", quality=90)
        jpegs.append(buf.getvalue())
    write_mjpeg_avi(os.path.join(output_dir, filename), jpegs, size[0], size[1])


def _load_rgb8(path, device):
    """Image.open(path) as the H x W x 3 uint8 device tensor the ingest kernels take (CenterPad indexes axis 2 of
    np.array(image), utils/util_distortion.py:231, so the reference needs three channels too)."""
    from PIL import Image
    arr = np.array(Image.open(path))
    if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
        raise ValueError(f"{path}: expected an 8-bit RGB image (got array shape {arr.shape}, dtype {arr.dtype})")
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device)


def colorize_video(opt, input_path, reference_file, output_path, nonlocal_net, colornet, vggnet):
    """test.py:29-124."""
    from .frame import ClipColorizer
    # parameters for wls filter
    wls_filter_on = True
    lambda_value = 500
    sigma_color = 4

    # processing folders
    mkdir_if_not(output_path)
    print("processing the folder:", input_path)
    path, dirs, filenames = os.walk(input_path).__next__()
    filenames.sort(key=lambda f: int("".join(filter(str.isdigit, f) or -1)))

    # if frame propagation: use the first frame as reference; otherwise, use the specified reference image
    # (string concatenation, not os.path.join, as upstream: --clip_path without a trailing slash only works
    # without --frame_propagate)
    ref_name = input_path + filenames[0] if opt.frame_propagate else reference_file
    print("reference name:", ref_name)

    device = torch.device("cuda", torch.cuda.current_device())
    image_size = opt.image_size
    cc = ClipColorizer(vggnet, nonlocal_net, colornet, temperature=1e-10)
    frame_ref = None if opt.frame_propagate else _load_rgb8(ref_name, device)
    if opt.frame_propagate:
        _load_rgb8(ref_name, device)          # upstream opens it (and fails here if it is missing)
    batch = max(1, int(getattr(opt, "batch_frames", 32)))
    index = 0
    with torch.no_grad():
        for b0 in range(0, len(filenames), batch):
            frames = [_load_rgb8(os.path.join(input_path, f), device) for f in filenames[b0:b0 + batch]]
            rgbs = cc.colorize_video(frames, frame_ref, image_size=image_size, wls_filter_on=wls_filter_on,
                                     lambda_value=lambda_value, sigma_color=sigma_color,
                                     frame_propagate=bool(opt.frame_propagate), continue_clip=b0 > 0)
            for IA_predict_rgb in rgbs:
                save_frames(IA_predict_rgb.cpu().numpy(), output_path, index)       # save the frames
                index += 1
    # output video
    folder2vid(image_folder=output_path, output_dir=output_path, filename="video.avi")
    print()


def colorize_video_refs(opt, input_path, reference_files, output_paths, nonlocal_net, colornet, vggnet):
    """`colorize_video` for all reference images of a clip in ONE pass over the frames: what test.py:169-181's loop
    `for ref_name in refs: colorize_video(...)` produces — folder `output_paths[r]` with `00000.jpg ...` and `video.avi` for
    reference `reference_files[r]` — with the frame decode / ingest and the frame-side front end done once per frame instead of
    once per (frame, reference).  With --frame_propagate the reference file is ignored upstream (test.py:50): the R folders
    receive the same frames, computed once."""
    from .frame import ClipColorizer
    wls_filter_on, lambda_value, sigma_color = True, 500, 4
    for o in output_paths:
        mkdir_if_not(o)
    print("processing the folder:", input_path)
    path, dirs, filenames = os.walk(input_path).__next__()
    filenames.sort(key=lambda f: int("".join(filter(str.isdigit, f) or -1)))
    device = torch.device("cuda", torch.cuda.current_device())
    propagate = bool(opt.frame_propagate)
    for r in reference_files:
        print("reference name:", input_path + filenames[0] if propagate else r)
    refs = None if propagate else [_load_rgb8(r, device) for r in reference_files]
    if propagate:
        _load_rgb8(input_path + filenames[0], device)
    cc = ClipColorizer(vggnet, nonlocal_net, colornet, temperature=1e-10)
    batch = max(1, int(getattr(opt, "batch_frames", 32)))
    index = 0
    with torch.no_grad():
        for b0 in range(0, len(filenames), batch):
            frames = [_load_rgb8(os.path.join(input_path, f), device) for f in filenames[b0:b0 + batch]]
            rgbs = cc.colorize_video(frames, refs if (refs is None or len(refs) > 1) else refs[0], image_size=opt.image_size,
                                     wls_filter_on=wls_filter_on, lambda_value=lambda_value, sigma_color=sigma_color,
                                     frame_propagate=propagate, continue_clip=b0 > 0)
            for per_ref in rgbs:
                images = per_ref if isinstance(per_ref, list) else [per_ref] * len(output_paths)
                for image, o in zip(images, output_paths):
                    save_frames(image.cpu().numpy(), o, index)
                index += 1
    for o in output_paths:
        folder2vid(image_folder=o, output_dir=o, filename="video.avi")
    print()


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--frame_propagate", default=False, type=bool, help="propagation mode, , please check the paper")
    parser.add_argument("--image_size", type=int, default=[216 * 2, 384 * 2], help="the image size, eg. [216,384]")
    parser.add_argument("--cuda", action="store_false")
    parser.add_argument("--gpu_ids", type=str, default="0", help="separate by comma")
    parser.add_argument("--clip_path", type=str, default="./sample_videos/clips/v32", help="path of input clips")
    parser.add_argument("--ref_path", type=str, default="./sample_videos/ref/v32", help="path of refernce images")
    parser.add_argument("--output_path", type=str, default="./sample_videos/output", help="path of output clips")
    # not upstream (all optional)
    parser.add_argument("--vgg_path", type=str, default="data/vgg19_conv.pth")
    parser.add_argument("--nonlocal_path", type=str, default=os.path.join("checkpoints/", "video_moredata_l1/nonlocal_net_iter_76000.pth"))
    parser.add_argument("--colornet_path", type=str, default=os.path.join("checkpoints/", "video_moredata_l1/colornet_iter_76000.pth"))
    parser.add_argument("--synthetic_weights", action="store_true", help="deterministic synthetic weights instead of checkpoints")
    parser.add_argument("--batch_frames", type=int, default=32, help="frames decoded and colourised per device batch")
    parser.add_argument("--refs_per_pass", type=int, default=1,
                        help="reference images colourised in one pass over the clip.  1 (default) = one pass per reference, the "
                             "upstream loop (test.py:169-181), bit-identical per reference; R > 1 = R references per pass "
                             "(1.46x the frame-colourisations/s at R = 4): ColorVidNet then runs at batch R under the batch-aware "
                             "launch plan and the saved frames agree with the per-reference runs to fp32 rounding of the "
                             "convolutions' summation order, not bit for bit")
    return parser


def main(argv=None):
    """test.py:126-186."""
    from .nets import ColorVidNet, VGG19_pytorch, WarpNet
    opt = build_parser().parse_args(argv)
    opt.gpu_ids = [int(x) for x in opt.gpu_ids.split(",")]
    print("running on GPU", opt.gpu_ids)
    torch.cuda.set_device(0)                      # test.py:24-26

    clip_name = opt.clip_path.split("/")[-1]
    refs = os.listdir(opt.ref_path)
    refs.sort()

    nonlocal_net = WarpNet(1)
    colornet = ColorVidNet(7)
    vggnet = VGG19_pytorch()
    if opt.synthetic_weights:
        from . import synth
        vggnet.load_state_dict(synth.vgg19_state_dict(0))
        nonlocal_net.load_state_dict(synth.warpnet_state_dict(0))
        colornet.load_state_dict(synth.colorvidnet_state_dict(0, contractive=True))
        print("using deterministic synthetic weights (no checkpoints)")
    else:
        vggnet.load_state_dict(torch.load(opt.vgg_path))
        print("succesfully load nonlocal model: ", opt.nonlocal_path)
        print("succesfully load color model: ", opt.colornet_path)
        nonlocal_net.load_state_dict(torch.load(opt.nonlocal_path))
        colornet.load_state_dict(torch.load(opt.colornet_path))
    for param in vggnet.parameters():
        param.requires_grad = False

    nonlocal_net.eval()
    colornet.eval()
    vggnet.eval()
    nonlocal_net.cuda()
    colornet.cuda()
    vggnet.cuda()

    per_pass = max(1, int(opt.refs_per_pass))
    out_of = lambda ref_name: os.path.join(opt.output_path, clip_name + "_" + ref_name.split(".")[0])     # noqa: E731
    for g0 in range(0, len(refs), per_pass):
        group = refs[g0:g0 + per_pass]
        if len(group) > 1:
            # all references of the group in one pass over the clip; a reference that cannot be colourised (unreadable
            # image, ...) must not take the others down with it, so a failed group falls back to the upstream loop
            try:
                colorize_video_refs(opt, opt.clip_path, [os.path.join(opt.ref_path, r) for r in group],
                                    [out_of(r) for r in group], nonlocal_net, colornet, vggnet)
                continue
            except Exception as error:
                print("error when colorizing the video with references " + ", ".join(group) + " in one pass; one by one:")
                print(error)
        for ref_name in group:
            try:
                colorize_video(opt, opt.clip_path, os.path.join(opt.ref_path, ref_name), out_of(ref_name),
                               nonlocal_net, colornet, vggnet)
            except Exception as error:
                print("error when colorizing the video " + ref_name)
                print(error)

    video_name = "video.avi"
    clip_output_path = os.path.join(opt.output_path, clip_name)
    mkdir_if_not(clip_output_path)
    folder2vid(image_folder=opt.clip_path, output_dir=clip_output_path, filename=video_name)


if __name__ == "__main__":
    main()
