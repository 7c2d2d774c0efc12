"""The only mounted interface of the reference is the two shell blocks of its README
(/root/reference/README.md:54-63 train, :68-75 test).  Their literal argument lists must parse
with the re-hosted tools (no GPU needed: parsers only)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "det", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# README.md:54-63, token for token (including the stray space in `-- rsu [0/1]`)
README_TRAIN = ["--data", "/path/to/training/dataset", "--com", "disco", "--log", "--batch", "4",
                "--kd_flag", "1", "--resume_teacher", "/path/to/teacher/checkpoint.pth",
                "--auto_resume_path", "logs", "--logpath", "logs", "--nepoch", "100", "--", "rsu", "[0/1]"]
# README.md:68-75
README_TEST = ["--data", "/path/to/testing/dataset", "--com", "disco", "--resume",
               "/path/to/teacher/checkpoint.pth", "--tracking", "--logpath", "logs", "--visualization", "1",
               "--rsu", "1"]


def test_readme_train_command_parses():
    t = _load("train_codet")
    a = t.parse_args(README_TRAIN)
    assert (a.com, a.batch, a.kd_flag, a.nepoch, a.log) == ("disco", 4, 1, 100, True)
    assert a.resume_teacher.endswith("checkpoint.pth") and a.auto_resume_path == "logs" and a.logpath == "logs"
    assert a.rsu == 1
    # the intended spelling works too
    assert t.parse_args(README_TRAIN[:-3] + ["--rsu", "0"]).rsu == 0


def test_readme_test_command_parses():
    t = _load("test_codet")
    a = t.build_parser().parse_args(README_TEST)
    assert a.com == "disco" and a.tracking and a.visualization == 1 and a.rsu == 1 and a.logpath == "logs"
    assert a.resume.endswith("checkpoint.pth")


def test_auto_resume_picks_the_newest_epoch(tmp_path):
    t = _load("train_codet")
    assert t.newest_checkpoint(str(tmp_path)) is None
    for n in (1, 12, 3):
        (tmp_path / ("epoch_%d.pth" % n)).write_bytes(b"x")
    (tmp_path / "epoch_final.pth").write_bytes(b"x")
    assert t.newest_checkpoint(str(tmp_path)).endswith("epoch_12.pth")
