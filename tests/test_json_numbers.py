"""include/MeshFEMHip/Json.hh accepts exactly the JSON number grammar: compiled and run as a small C++ program."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_json_number_grammar(tmp_path):
    exe = str(tmp_path / "json_numbers")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "json_numbers.cc"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert "REJECTED" not in out and "ACCEPTED" not in out, out
    assert out.count("ok ") == 5 and out.count("rejected ") == 11 and out.strip().endswith("1.5 -25"), out
