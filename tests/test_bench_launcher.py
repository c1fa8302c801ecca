"""`python bench.py --gpus N` without a launcher must produce N ranks by itself (the reference starts its ranks with torchrun,
ppo.sh:59): the parent -- which touches no GPU -- starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child, relays rank 0's JSON line and fails loudly if a rank fails or the line does not report N ranks.  CPU: the launcher logic
is exercised with stand-in rank scripts (the real ranks need a HIP device)."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launcher_command_is_one_process_per_gpu_on_loopback():
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "5"], "/x/bench.py", 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]


def test_self_launch_starts_n_ranks_and_relays_rank0_line(tmp_path, capfd):
    s = _script(tmp_path, """
        import json, os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")
        assert os.environ["LR2_BENCH_CHILD"] == "1" and sys.argv[1:] == ["--gpus", "2", "--steps", "3"]
        os.write(1, ("chatter from rank %d" % dist.get_rank() + os.linesep).encode())      # ONE write per line: two ranks share the pipe
        if dist.get_rank() == 0:
            os.write(1, (json.dumps({"metric": "ppo_steps_per_sec", "n_gpus": dist.get_world_size(), "value": 1.0}) + os.linesep).encode())
        dist.destroy_process_group()
    """)
    assert bench.self_launch(2, ["--gpus", "2", "--steps", "3"], script=s) == 0
    out, err = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2          # stdout: exactly the JSON line
    assert "chatter from rank 0" in err and "chatter from rank 1" in err


def test_self_launch_fails_when_a_rank_fails_or_the_line_reports_fewer_ranks(tmp_path, capfd):
    bad_rank = _script(tmp_path, """
        import os, sys
        if os.environ["RANK"] == "1":
            sys.exit(7)
        print('{"metric": "ppo_steps_per_sec", "n_gpus": 2}')
    """)
    assert bench.self_launch(2, [], script=bad_rank) != 0
    one_rank_line = _script(tmp_path, """
        import os
        if os.environ["RANK"] == "0":
            print('{"metric": "ppo_steps_per_sec", "n_gpus": 1}')
    """)
    assert bench.self_launch(2, [], script=one_rank_line) == 3
    silent = _script(tmp_path, "pass\n")
    assert bench.self_launch(2, [], script=silent) == 3
    out, _ = capfd.readouterr()
    assert out.strip() == ""                                                # nothing that looks like a result was passed on


def test_bench_gpus_2_on_a_box_without_gpu_exits_nonzero_instead_of_reporting_one_rank():
    """The round-3 behaviour was a stderr note and a ONE-rank benchmark; now `--gpus 2` either runs 2 ranks or fails."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    if r.returncode == 0:            # a GPU box: then the line must say 2
        assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["n_gpus"] == 2
    else:
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
