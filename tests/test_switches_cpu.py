"""The switches of the library and of the command line are lists, not habits (round 6):
  * libcolorid_hip.so reads its environment in ONE function (cid_api_ctx.hip: read_switches, once per context) — no other getenv in csrc/;
  * every variable it reads is a row of cid_switches.def, and cid_tunables / cid_ctx_tune are generated from the same rows;
  * the command line asks cli_env("NAME") only, every NAME is a row of host/cli_switches.def;
  * README.md's table is the generator's output over those two files."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "colorid_amd", "csrc")


def _rows(path, macro):
    return re.findall(r'^%s\((\w+),\s*"(\w+)",\s*(\w),\s*([-\w]+),\s*"(.*)"\)\s*$' % macro, open(path).read(), re.M)


def test_library_reads_its_environment_in_one_place():
    offenders = []
    for f in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")):
        text = re.sub(r"//.*", "", open(f).read())
        n = len(re.findall(r"\bgetenv\s*\(", text))
        if n and os.path.basename(f) != "cid_api_ctx.hip":
            offenders.append((os.path.basename(f), n))
    assert offenders == []
    ctx = open(os.path.join(CSRC, "cid_api_ctx.hip")).read()
    body = ctx[ctx.index("static void read_switches("):ctx.index("int cid_ctx_create(")]
    outside = re.sub(r"//.*", "", ctx.replace(body, ""))
    assert not re.findall(r"\bgetenv\s*\(", outside)
    listed = {env for _, env, _, _, _ in _rows(os.path.join(CSRC, "cid_switches.def"), "CID_SWITCH")}
    assert set(re.findall(r'getenv\("(\w+)"\)', body)) <= listed          # the words (COLORID_REDUCE ...) are rows too
    assert len(listed) >= 25 and len(listed) == len(_rows(os.path.join(CSRC, "cid_switches.def"), "CID_SWITCH"))


def test_command_line_asks_cli_env_only():
    listed = {env for _, env, _, _, _ in _rows(os.path.join(CSRC, "host", "cli_switches.def"), "CLI_SWITCH")}
    used = set()
    for f in glob.glob(os.path.join(CSRC, "host", "*.cpp")) + glob.glob(os.path.join(CSRC, "host", "*.hpp")):
        text = open(f).read()
        used |= set(re.findall(r'cli_env\("(\w+)"\)', re.sub(r"//.*", "", text)))
        for line in re.sub(r"//.*", "", text).splitlines():
            if re.search(r"\bgetenv\s*\(", line):
                assert "getenv(env)" in line or "getenv(v)" in line, (os.path.basename(f), line.strip()[:120])   # cli_env's own snapshot; main.cpp's list of OTHER tools' variables
    assert used <= listed, used - listed
    assert listed - used == set(), listed - used


def test_readme_table_is_generated():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_switch_table.py"), "--check"])
    assert p.returncode == 0, "README.md's switch table is stale: run python3 tools/gen_switch_table.py"
    readme = open(os.path.join(ROOT, "README.md")).read()
    for _, env, _, _, _ in _rows(os.path.join(CSRC, "cid_switches.def"), "CID_SWITCH") + _rows(os.path.join(CSRC, "host", "cli_switches.def"), "CLI_SWITCH"):
        assert "`%s`" % env in readme
