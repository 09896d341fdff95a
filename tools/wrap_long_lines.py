#!/usr/bin/env python3
"""Re-wrap over-long C++/HIP source lines (no clang-format in the image): a trailing // comment moves onto its own line above, code
breaks after a `, ` / ` && ` / ` || ` / ` ? ` / ` : ` / ` << ` that sits outside string literals, continuation indented by 4 (aligned
with the opening parenthesis where that leaves room).  Preprocessor lines and lines inside block comments are left alone.
usage: wrap_long_lines.py [--limit 150] [--over 160] files..."""
import re
import sys


def split_code_comment(line):
    """-> (code, comment) where comment starts at a // outside string / char literals (or None)"""
    i, n, q = 0, len(line), None
    while i < n:
        c = line[i]
        if q:
            if c == "\\":
                i += 2
                continue
            if c == q:
                q = None
        elif c in "\"'":
            q = c
        elif c == "/" and i + 1 < n and line[i + 1] == "/":
            return line[:i].rstrip(), line[i:]
        i += 1
    return line, None


def break_points(code):
    """indices AFTER which the line may break (outside literals), with the parenthesis depth there"""
    pts, q, depth, i, n = [], None, 0, 0, len(code)
    stack = []
    while i < n:
        c = code[i]
        if q:
            if c == "\\":
                i += 2
                continue
            if c == q:
                q = None
        elif c in "\"'":
            q = c
        elif c in "([{":
            stack.append(i)
        elif c in ")]}":
            if stack:
                stack.pop()
        else:
            for tok in (", ", " && ", " || ", " ? ", " : ", " << ", "; "):
                if code.startswith(tok, i):
                    end = i + len(tok)
                    if tok in (" && ", " || ", " ? ", " : ", " << "):   # operators start the continuation line
                        pts.append((i + 1, list(stack)))
                    else:
                        pts.append((end, list(stack)))
                    break
        i += 1
    return pts


def wrap(line, limit):
    indent = len(line) - len(line.lstrip(" "))
    code, comment = split_code_comment(line)
    out = []
    if comment is not None and code.strip():
        out.append(" " * indent + comment)
        line = code
    elif comment is not None:
        return [line]            # a pure comment line: leave prose alone
    else:
        line = code
    while len(line) > limit:
        pts = [p for p in break_points(line) if indent + 8 < p[0] <= limit]
        if not pts:
            break
        at, stack = pts[-1]
        cont = indent + 4
        if stack and stack[-1] + 1 <= limit // 2:
            cont = stack[-1] + 1
        out.append(line[:at].rstrip())
        line = " " * cont + line[at:].lstrip()
        indent = cont - 4 if cont >= 4 else 0
    out.append(line)
    return out


def main():
    args = sys.argv[1:]
    limit, over = 150, 160
    while args and args[0].startswith("--"):
        if args[0] == "--limit":
            limit = int(args[1])
        elif args[0] == "--over":
            over = int(args[1])
        args = args[2:]
    for path in args:
        src = open(path).read().split("\n")
        out, in_block, changed = [], False, 0
        for ln in src:
            stripped = ln.lstrip()
            if in_block or stripped.startswith("#") or ln.rstrip().endswith("\\") or len(ln) <= over:
                out.append(ln)
            else:
                w = wrap(ln, limit)
                changed += w != [ln]
                out.extend(w)
            if "/*" in ln and "*/" not in ln.split("/*")[-1]:
                in_block = True
            elif in_block and "*/" in ln:
                in_block = False
        if changed:
            open(path, "w").write("\n".join(out))
        print(f"{path}: {changed} line(s) re-wrapped")


if __name__ == "__main__":
    main()
