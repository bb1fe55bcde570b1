"""sample/sample.py of the reference on the MI355X engine: single process, one video.

  python tools/sample.py --config /path/to/configs/ffs/ffs_sample.yaml [--ckpt model.pt] [--vae DIR] [--steps N]

Same flow as /root/reference/sample/sample.py:39-126 with the three import lines swapped."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

if __name__ == "__main__":
    import sample_ddp
    os.environ.setdefault("WORLD_SIZE", "1")
    if "--num-samples" not in sys.argv:
        sys.argv += ["--num-samples", "1"]
    sample_ddp.main()
