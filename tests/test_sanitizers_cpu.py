"""Address/UB-sanitizer run of the plain-C host code and the oracle (CPU build only — GPU sanitizers are not
available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_c_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_host")
    src = [os.path.join(ROOT, "tests", "asan_host.c"),
           os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "bsw_refbatch.c"),
           os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "bsw_glue.c"),
           os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "bsw_synth.c"),
           os.path.join(ROOT, "oracle", "ksw_extend_ref.c"), os.path.join(ROOT, "oracle", "rowsync_model.c")]
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "include"), "-o", exe] + src + ["-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    assert "asan_host ok" in out.stdout
