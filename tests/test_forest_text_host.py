"""CPU: forest-em's text formats in the front end (carmel_amd/csrc/host/forest_text.hpp), through a small compiled driver."""
import os
import subprocess

import pytest

from conftest import ROOT

DRIVER = r'''
#include <cmath>
#include <cstdio>
#include <iostream>
#include <limits>
#include "forest_text.hpp"
using namespace carmel_host;
int main() {
  const double inf = std::numeric_limits<double>::infinity();
  const double lw[5] = {std::log(0.5), -300.0, -inf, 0.0, std::log(0.125)};
  const std::string sometimes = write_params(lw, 5, W_SOMETIMES_LOG), never = write_params(lw, 5, W_NEVER_LOG);
  std::cout << sometimes << "--\n" << never << "--\n";
  for (const std::string& text : {sometimes, never, std::string("(0.5 e^-300 0 1 .125)\n"), std::string("0.5, e^-300\n0\n1 .125")}) {
    std::vector<double> back = read_params(text);
    std::printf("%zu", back.size());
    for (double v : back) std::printf(" %.17g", v);
    std::printf("\n");
  }
  // forest-em/sample/norm_and_forests: the groups' list is followed by forests, which are not its business
  std::vector<uint64_t> off;
  std::vector<uint32_t> rule;
  uint32_t mx = 0;
  read_normgroups("((1 2 7 ) (3 4 5 6))\n(1 4)\n(OR (1 4) (1 3))\n", off, rule, mx);
  std::printf("groups %zu rules %zu max %u\n", off.size() - 1, rule.size(), mx);
  for (const char* bad : {"", "   ", "((1 2) (3", "(OR 1 2)"}) {
    try {
      read_normgroups(bad, off, rule, mx);
      std::printf("accepted\n");
    } catch (std::exception& e) {
      std::printf("refused: %s\n", e.what());
    }
  }
  return 0;
}
'''


def test_parameter_files_are_written_and_read_as_the_reference_does(tmp_path):
    """FForests::write_params / write_counts (forest-em.hpp:190-201: print_range multiline, no parentheses, then endl --
    graehl/shared/io.hpp:327-343): a space, the weight and a newline per parameter and an empty line at the end, the format of
    forest-em/sample/best_weights; read_params takes that, the parenthesised vector and comma-separated weights alike"""
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    inc = os.path.join(ROOT, "carmel_amd", "csrc", "host")
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-I", inc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    if r.returncode != 0:
        pytest.fail(r.stderr[-2000:])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, universal_newlines=True).stdout
    out, groups = out.split("groups ", 1)
    assert groups.split("\n")[0] == "2 rules 7 max 7"
    assert groups.split("\n")[1:5] == ["refused: Expected normalization groups list e.g. ((1 2 3) (4 5) (6))"] * 2 + \
        ["refused: normalisation groups: unbalanced parentheses", "refused: normalisation groups: unexpected character 'O'"]
    sometimes, never, rest = out.split("--\n")
    assert sometimes == " 0.5\n e^-300\n 0\n 1\n 0.125\n\n"
    assert never.split("\n")[0] == " 0.5" and never.endswith("\n\n") and "e^" not in never
    rows = [l.split() for l in rest.strip().split("\n")]
    assert [r[0] for r in rows] == ["5", "5", "5", "5"]
    import math
    for i in (0, 2, 3):  # e^-300 survives in the log forms; the never-log file prints it as a real number
        vals = [float(x) for x in rows[i][1:]]
        assert vals[0] == pytest.approx(math.log(0.5)) and vals[1] == pytest.approx(-300.0) and vals[2] == -math.inf
        assert vals[3] == 0.0 and vals[4] == pytest.approx(math.log(0.125))
    assert float(rows[1][2]) == pytest.approx(-300.0, rel=1e-12)


def test_forest_em_refuses_what_the_reference_refuses_before_any_device_call(tmp_path):
    """forest-em-params.cpp:55-60: an annotated rule file without the rule file, EM without normalisation groups -- and no device is
    asked for before the complaint (there is none here); unknown switches are refused too"""
    import subprocess
    from conftest import GOLDEN, ROOT
    cli = os.path.join(ROOT, "carmel_amd", "bin", "forest-em")
    if not os.path.exists(cli):
        pytest.skip("front end not built")
    f, n = os.path.join(GOLDEN, "fem.forests"), os.path.join(GOLDEN, "fem.norm")
    run = lambda args: subprocess.run([cli] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    p = run(["-f", f, "-n", n, "-i", "1", "-B", str(tmp_path / "out")])
    assert p.returncode != 0 and "Must provide byid-rule-file." in p.stderr and not os.path.exists(tmp_path / "out")
    p = run(["-f", f, "-i", "1"])
    assert p.returncode != 0 and "Missing normgroups-file." in p.stderr
    p = run(["-f", f, "-n", n, "--no-such-switch"])
    assert p.returncode != 0 and "unknown option" in p.stderr
