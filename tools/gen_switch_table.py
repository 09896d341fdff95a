#!/usr/bin/env python3
"""README.md's table of switches, generated: the library's from colorid_amd/csrc/cid_switches.def (the one list its code is built from),
the command line's from colorid_amd/csrc/host/cli_switches.def.  `python3 tools/gen_switch_table.py` rewrites the block between the
markers in README.md; `--check` exits 1 when the block is stale (tests/test_switches_cpu.py)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- switches:begin (tools/gen_switch_table.py) -->", "<!-- switches:end -->"


def parse(path, macro):
    rows = []
    for m in re.finditer(r'^%s\((\w+),\s*"(\w+)",\s*(\w),\s*([-\w]+),\s*"(.*)"\)\s*$' % macro, open(path).read(), re.M):
        rows.append(m.groups())
    return rows


def table():
    out = [BEGIN, "", "**Library** (`libcolorid_hip.so`; read once when a context is made, or `cid_ctx_tune(ctx, name, value)` per context):", "",
           "| environment variable | `cid_ctx_tune` name | default | what it does |", "|---|---|---|---|"]
    for name, env, kind, dflt, doc in parse(os.path.join(ROOT, "colorid_amd", "csrc", "cid_switches.def"), "CID_SWITCH"):
        out.append(f"| `{env}` | `{name}` | {dflt} | {doc} |")
    out += ["", "**Command line** (`colorid`; read by `cli_env()` in `csrc/host/`):", "", "| environment variable | default | what it does |", "|---|---|---|"]
    for name, env, kind, dflt, doc in parse(os.path.join(ROOT, "colorid_amd", "csrc", "host", "cli_switches.def"), "CLI_SWITCH"):
        out.append(f"| `{env}` | {dflt} | {doc} |")
    out += ["", END]
    return "\n".join(out)


def main():
    p = os.path.join(ROOT, "README.md")
    s = open(p).read()
    new = table()
    if BEGIN in s:
        cur = s[s.index(BEGIN):s.index(END) + len(END)]
        if "--check" in sys.argv:
            sys.exit(0 if cur == new else 1)
        s = s.replace(cur, new)
    else:
        if "--check" in sys.argv:
            sys.exit(1)
        s = s.rstrip("\n") + "\n\n## Switches\n\n" + new + "\n"
    open(p, "w").write(s)


if __name__ == "__main__":
    main()
