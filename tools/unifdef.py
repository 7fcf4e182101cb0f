#!/usr/bin/env python3
"""tools/unifdef.py FILE... -D NAME=VALUE ...  — remove preprocessor branches that the given macro values decide.

A conditional whose outcome does not depend on the macros that stay undefined is resolved (the taken branch is kept, the
directives and the other branches go); one that still depends on an unknown macro is left alone and reported.  Used in
round 4 to take the measured-and-lost experiments (HR_STEAL, HR_NODE32, ...) out of the product sources; the patches that
put them back are under profiles/experiments/.
"""
import itertools
import re
import sys


def decide(expr, known):
    """True / False when the known macros decide `expr`, None otherwise."""
    expr = re.sub(r"//.*$", "", expr).strip()
    expr = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if m.group(1) in known else "__U_" + m.group(1), expr)
    names = sorted(set(re.findall(r"[A-Za-z_]\w*", expr)))
    unknown = [n for n in names if n not in known]
    if len(unknown) > 4:
        return None
    py = expr.replace("&&", " and ").replace("||", " or ")
    py = re.sub(r"!(?!=)", " not ", py)
    results = set()
    for combo in itertools.product((0, 1, 2), repeat=len(unknown)):
        env = dict(known)
        env.update(dict(zip(unknown, combo)))
        try:
            results.add(bool(eval(py, {"__builtins__": {}}, env)))
        except Exception:
            return None
    return results.pop() if len(results) == 1 else None


def process(text, known, path):
    out = []
    # stack entries: [state, taken] with state in {"keep" (undecided: directives stay), "on", "off"}; taken: a decided branch was kept
    stack = []

    def emitting():
        return all(s[0] != "off" for s in stack)

    for ln, line in enumerate(text.split("\n"), 1):
        s = line.strip()
        m = re.match(r"#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting():
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            if not emitting():
                stack.append(["off", True, "dead"])
                continue
            if kind == "if":
                d = decide(rest, known)
            else:
                name = rest.strip().split()[0]
                d = None
                if name in known:
                    d = kind == "ifdef"
            if d is None:
                if any(k in rest for k in known):
                    print(f"{path}:{ln}: left alone: {s}", file=sys.stderr)
                stack.append(["keep", False, "keep"])
                out.append(line)
            else:
                stack.append(["on" if d else "off", d, "decided"])
        elif kind == "elif":
            top = stack[-1]
            if top[2] == "dead":
                continue
            if top[2] == "keep":
                out.append(line)
                continue
            if top[1]:
                top[0] = "off"
            else:
                d = decide(rest, known)
                if d is None:
                    raise SystemExit(f"{path}:{ln}: #elif after a decided #if depends on unknown macros: {s}")
                top[0] = "on" if d else "off"
                top[1] = d
        elif kind == "else":
            top = stack[-1]
            if top[2] == "dead":
                continue
            if top[2] == "keep":
                out.append(line)
                continue
            top[0] = "off" if top[1] else "on"
            top[1] = True
        else:  # endif
            top = stack.pop()
            if top[2] == "keep":
                out.append(line)
    if stack:
        raise SystemExit(f"{path}: unbalanced conditionals")
    return "\n".join(out)


def main():
    files, known = [], {}
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "-D":
            k, _, v = args.pop(0).partition("=")
            known[k] = int(v or "1")
        else:
            files.append(a)
    for f in files:
        src = open(f).read()
        dst = process(src, known, f)
        if dst != src:
            open(f, "w").write(dst)
            print(f"{f}: {len(src.splitlines())} -> {len(dst.splitlines())} lines", file=sys.stderr)


if __name__ == "__main__":
    main()
