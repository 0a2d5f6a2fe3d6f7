"""Same-box A/B of two builds of libtgs_hip.so on the WHOLE train step (bench.py's timed region):

    python tools/ab_step.py A.so B.so [rounds] [bench args...]

alternates the libraries (TGS_LIB_PATH), one bench.py process each, and prints ms_per_step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    extra = sys.argv[4:]
    for r in range(rounds):
        for tag, lib in (("A", a), ("B", b)):
            env = dict(os.environ, TGS_LIB_PATH=os.path.abspath(lib))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "50",
                                  "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, out.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            print(tag, os.path.basename(lib), d["ms_per_step"], "ms/step", d["value"], "it/s  asis", d.get("value_asis_layout"),
                  d["kernel_ms"], flush=True)


if __name__ == "__main__":
    main()
