"""Drop-in for /root/reference/test.py: same command line, same `colorize_video` signature, the MI355X hot path
underneath (see dvc_amd/cli.py for what is kept, including the upstream flag quirks, and what moved to the device).

    python test.py --clip_path ./sample_videos/clips/v32 --ref_path ./sample_videos/ref/v32 --output_path ./out
"""
from dvc_amd.cli import colorize_video, folder2vid, main, mkdir_if_not, save_frames  # noqa: F401

if __name__ == "__main__":
    main()
