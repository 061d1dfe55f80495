#!/usr/bin/env python3
"""Resolve the preprocessor conditionals of a source file that depend ONLY on a given set of macro values (an `unifdef` for
expressions like `#if T9_DIET & 16`): the taken branch stays, the others and the directives go; conditionals on anything else
are left alone.  Used at the end of round 6 to freeze the ablation switches of msda_tiled9.hip / msda_bwd_mfma.hip at their
defaults (the losing sides are in the git history).
    python tools/freeze_ifdefs.py file.hip NAME=VALUE ... [-U UNDEFINED_NAME ...]"""
import re
import sys


def main():
    path = sys.argv[1]
    vals, undef = {}, set()
    it = iter(sys.argv[2:])
    for a in it:
        if a == "-U":
            undef.add(next(it))
        else:
            k, v = a.split("=")
            vals[k] = int(v)
    known = set(vals) | undef

    def evaluate(expr):
        """-> True / False, or None when the expression mentions anything we do not know"""
        e = re.sub(r"//.*$", "", expr).strip()
        e = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", lambda m: f" __DEF_{m.group(1) or m.group(2)}__ ", e)
        names = set(re.findall(r"[A-Za-z_]\w*", e))
        for n in names:
            if n.startswith("__DEF_"):
                if n[6:-2] not in known:
                    return None
            elif n not in vals:
                return None
        for n in sorted(names, key=len, reverse=True):
            if n.startswith("__DEF_"):
                e = e.replace(n, "1" if n[6:-2] in vals else "0")
            else:
                e = re.sub(rf"\b{n}\b", str(vals[n]), e)
        e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
        return bool(eval(e, {"__builtins__": {}}))

    out = []
    stack = []   # entries: dict(mode='resolved'|'kept', taken=bool (a branch already taken), active=bool (emit lines now))
    for line in open(path).read().split("\n"):
        s = line.strip()
        m = re.match(r"#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", s)
        emitting = all(f["active"] for f in stack)
        if not m:
            if emitting:
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("if", "ifdef", "ifndef"):
            if not emitting:
                stack.append(dict(mode="dead", taken=True, active=False))
                continue
            if d == "if":
                v = evaluate(rest)
            else:
                name = re.sub(r"//.*$", "", rest).strip()
                v = None if name not in known else ((name in vals) == (d == "ifdef"))
            if v is None:
                stack.append(dict(mode="kept", taken=True, active=True))
                out.append(line)
            else:
                stack.append(dict(mode="resolved", taken=v, active=v))
        elif d == "elif":
            f = stack[-1]
            if f["mode"] == "kept":
                out.append(line)
            elif f["mode"] == "resolved":
                if f["taken"]:
                    f["active"] = False
                else:
                    v = evaluate(rest)
                    assert v is not None, f"#elif on unknown names inside a resolved conditional: {line}"
                    f["taken"] = f["active"] = v
        elif d == "else":
            f = stack[-1]
            if f["mode"] == "kept":
                out.append(line)
            elif f["mode"] == "resolved":
                f["active"] = not f["taken"]
                f["taken"] = True
        else:
            f = stack.pop()
            if f["mode"] == "kept":
                out.append(line)
    assert not stack
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
