"""The nerfstudio registration of pointnerf2studio_amd/studio_config.py (TrainerConfig, MethodSpecification, the
datamanager and pipeline shells: studio_config.py:14-50, studio_pipeline.py:16-53, studio_datamanager.py:41-60 of the
reference) executed against a stand-in `nerfstudio` package -- see tests/fake_nerfstudio_check.py, which runs in its own
process because it rewires sys.modules."""
import os
import subprocess
import sys


def test_registration_block_runs_against_a_stand_in_nerfstudio():
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_nerfstudio_check.py")
    p = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "registration ok" in p.stdout
