"""Helper process of the GPU test run: started by conftest.py BEFORE anything touches the GPU, it runs commands on
request (one JSON line in, one JSON line out).  Multi-process tests (torch.distributed.run + bench.py) are started
through it because a process that has initialised the GPU must not fork/exec other programs on the GPU pool."""
import json
import os
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        env = dict(os.environ)
        for k in req.get("drop_env") or ():
            env.pop(k, None)
        env.update(req.get("env") or {})
        try:
            p = subprocess.run(req["cmd"], env=env, cwd=req.get("cwd"), capture_output=True, text=True, timeout=req.get("timeout", 600))
            out = {"rc": p.returncode, "stdout": p.stdout[-20000:], "stderr": p.stderr[-8000:]}
        except subprocess.TimeoutExpired as e:
            out = {"rc": -9, "stdout": (e.stdout or b"").decode(errors="replace")[-20000:] if isinstance(e.stdout, bytes) else (e.stdout or ""),
                   "stderr": "timeout"}
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
