#!/usr/bin/env python3
"""Public surface of TypeScript / JavaScript sources as DATA: class -> public method -> parameter counts, exported functions, interface
members.  Used two ways:

    python scripts/ts_surface.py --reference /root/reference  > tests/golden/reference_surface.json

extracts the surface of the reference's operator layer (src/renderers/*.ts, src/sort/sort_dynamic.ts, src/prefix/prefix.ts,
src/utils/allocate-pointcloud.ts, src/trainer.ts, and the loaders / camera / viewer files of SURVEY 8(f)) -- names and arities only, no source text -- into a committed fixture; and
tests/test_ts_surface.py parses bindings/ts/*.js and *.d.ts with the same scanner and checks that they cover that fixture.

The scanner is a brace-depth walk over comment- and string-stripped text, not a TypeScript parser: it understands what these files use
(class bodies with modifier keywords, one-line methods, `name(args): T {`, `name(args): T;` in .d.ts, interfaces with function-typed
members, `export function`)."""
from __future__ import annotations

import json
import os
import re
import sys

MODIFIERS = {"public", "private", "protected", "static", "async", "readonly", "override", "abstract", "declare", "export", "get", "set"}
NOT_METHODS = {"if", "for", "while", "switch", "catch", "return", "function", "new", "typeof", "await", "super", "throw"}

REFERENCE_FILES = ["src/renderers/tiled-forward-pass.ts", "src/renderers/tiled-rasterizer.ts", "src/renderers/tiled-backward-pass.ts", "src/renderers/optimizer.ts",
                   "src/renderers/densify-prune.ts", "src/sort/sort_dynamic.ts", "src/prefix/prefix.ts", "src/utils/allocate-pointcloud.ts", "src/trainer.ts",
                   # the callers and data formats either side of the path: loaders, image ingest, camera block, viewer
                   "src/viewer.ts", "src/camera/camera.ts", "src/utils/plyreader.ts", "src/utils/load-pointcloud.ts", "src/utils/load-camera.ts", "src/utils/load-images.ts"]


def strip(text: str) -> str:
    """Comments and string / template literals blanked out (same length, so offsets survive)."""
    out, i, n = list(text), 0, len(text)
    while i < n:
        c, d = text[i], text[i + 1] if i + 1 < n else ""
        if c == "/" and d == "/":
            j = text.find("\n", i)
            j = n if j < 0 else j
            out[i:j] = " " * (j - i); i = j
        elif c == "/" and d == "*":
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out[i:j] = [ch if ch == "\n" else " " for ch in text[i:j]]; i = j
        elif c in "'\"`":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out[i + 1:j] = [ch if ch == "\n" else " " for ch in text[i + 1:j]]; i = j + 1
        else:
            i += 1
    return "".join(out)


def matching(text: str, i: int) -> int:
    """Index of the bracket closing the one at text[i]."""
    pairs = {"(": ")", "{": "}", "[": "]"}
    open_c, close_c, depth = text[i], pairs[text[i]], 0
    for j in range(i, len(text)):
        if text[j] == open_c:
            depth += 1
        elif text[j] == close_c:
            depth -= 1
            if depth == 0:
                return j
    return len(text) - 1


def count_params(params: str) -> dict:
    parts, depth, angle, cur = [], 0, 0, ""
    for k, ch in enumerate(params):
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "<" and k > 0 and (params[k - 1].isalnum() or params[k - 1] == "_"):
            angle += 1
        elif ch == ">" and angle > 0 and params[k - 1] != "=":
            angle -= 1
        if ch == "," and depth == 0 and angle == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    parts = [p.strip() for p in parts if p.strip()]
    required = 0
    for p in parts:
        head = re.split(r"[:=]", p, 1)[0]
        top_default = re.search(r"(?<![=!<>])=(?![=>])", re.sub(r"\{[^{}]*\}|\[[^\[\]]*\]", "", p)) is not None
        if not head.rstrip().endswith("?") and not top_default and not p.startswith("..."):
            required += 1
    return dict(params=len(parts), required=required)


def members(body: str) -> dict:
    """Methods declared at the top level of a class / interface body: name -> {params, required, visibility}.  Fields -> {params: None}."""
    out, i, n, depth = {}, 0, len(body), 0
    stmt_start = 0
    while i < n:
        ch = body[i]
        if ch in "{[(":
            if depth == 0 and ch == "(":
                head = body[stmt_start:i]
                words = re.findall(r"[A-Za-z_$][\w$]*", head)
                close = matching(body, i)
                name = words[-1] if words else ""
                mods = set(words[:-1])
                arrow_field = re.search(r"[:=]\s*$", head) is not None      # `name: (args) => T` / `name = (args) => {`
                if arrow_field and words:
                    name, mods = words[0] if words[0] not in MODIFIERS else (words[1] if len(words) > 1 else ""), set(w for w in words if w in MODIFIERS)
                if name and name not in NOT_METHODS and (mods <= MODIFIERS or arrow_field) and not re.search(r"[=.]\s*\w*$", head.replace(name, "", 1) if not arrow_field else ""):
                    vis = "private" if ("private" in mods or "protected" in mods or name.startswith("_") or name.startswith("#")) else "public"
                    kind = "getter" if "get" in mods else ("setter" if "set" in mods else "method")
                    info = dict(count_params(body[i + 1:close]), visibility=vis, kind=kind, static="static" in mods)
                    out.setdefault(name, info)
                # skip to the end of this member: past the parameter list, an optional return type, and a body or a semicolon
                j = close + 1
                while j < n and body[j] not in "{;\n" or (j < n and body[j] == "\n" and re.match(r"\s*[:{=]", body[j:j + 40] or "")):
                    if body[j] in "([{":
                        j = matching(body, j)
                    j += 1
                if j < n and body[j] == "{":
                    j = matching(body, j)
                i = j + 1
                stmt_start = i
                continue
            if ch == "{":   # `name: { ... }` / `name = { ... }`: an object-typed or object-initialised field
                m = re.match(r"\s*((?:(?:public|private|protected|static|readonly|declare)\s+)*)([A-Za-z_$][\w$]*)\s*[?!]?\s*[:=]\s*$", body[stmt_start:i])
                if m and m.group(2) not in NOT_METHODS and m.group(2) not in MODIFIERS:
                    mods = set(m.group(1).split())
                    out.setdefault(m.group(2), dict(params=None, required=None, visibility="private" if ("private" in mods or "protected" in mods) else "public", kind="field",
                                                    static="static" in mods))
            i = matching(body, i) + 1
            if ch == "{":
                stmt_start = i
            continue
        if ch in ";,\n" and depth == 0:
            starts_member = ch == "\n" and re.match(r"\s*((public|private|protected|static|readonly|async|get|set|declare)\s+)*[A-Za-z_$][\w$]*\s*[?!]?\s*[:(=<]", body[i + 1:i + 200]) is not None
            if ch in ";," or starts_member or body[stmt_start:i].strip() == "" or re.search(r"[;,}]\s*$", body[stmt_start:i]):
                seg = body[stmt_start:i]
                m = re.match(r"\s*((?:(?:public|private|protected|static|readonly|declare)\s+)*)([A-Za-z_$][\w$]*)\s*[?!]?\s*[:=]", seg)
                if m and m.group(2) not in NOT_METHODS and m.group(2) not in MODIFIERS:
                    mods = set(m.group(1).split())
                    vis = "private" if ("private" in mods or "protected" in mods or m.group(2).startswith("_")) else "public"
                    out.setdefault(m.group(2), dict(params=None, required=None, visibility=vis, kind="field", static="static" in mods))
                stmt_start = i + 1
        i += 1
    return out


def surface(text: str) -> dict:
    s = strip(text)
    classes, interfaces, functions = {}, {}, {}
    for m in re.finditer(r"\b(export\s+)?(?:default\s+)?(?:abstract\s+)?(class|interface)\s+([A-Za-z_$][\w$]*)[^{;]*\{", s):
        open_i = m.end() - 1
        close_i = matching(s, open_i)
        mem = members(s[open_i + 1:close_i])
        (classes if m.group(2) == "class" else interfaces)[m.group(3)] = dict(exported=bool(m.group(1)), members=mem)
    for m in re.finditer(r"(^|\n)\s*(export\s+)?(?:declare\s+)?(?:async\s+)?function\s+([A-Za-z_$][\w$]*)\s*(<[^>(]*>)?\s*\(", s):
        open_i = m.end() - 1
        functions[m.group(3)] = dict(count_params(s[open_i + 1:matching(s, open_i)]), exported=bool(m.group(2)))
    # fields assigned in constructors: `this.name = ...` (JavaScript classes declare their fields that way)
    for cname, c in classes.items():
        m = re.search(r"\bclass\s+" + re.escape(cname) + r"\b[^{]*\{", s)
        body = s[m.end() - 1:matching(s, m.end() - 1)]
        for f in re.findall(r"\bthis\.([A-Za-z_$][\w$]*)\s*=", body):
            c["members"].setdefault(f, dict(params=None, required=None, visibility="private" if f.startswith("_") else "public", kind="field", static=False))
    return dict(classes=classes, interfaces=interfaces, functions=functions)


def reference_surface(root: str) -> dict:
    out = dict(source="names and arities extracted from the reference's TypeScript by scripts/ts_surface.py (no source text)", files={})
    for rel in REFERENCE_FILES:
        sf = surface(open(os.path.join(root, rel)).read())
        keep = dict(classes={k: {n: {a: b for a, b in v.items() if a != "visibility"} for n, v in c["members"].items() if v["visibility"] == "public" and v["kind"] != "field"}
                             for k, c in sf["classes"].items() if c["exported"]},
                    interfaces={k: {n: v["params"] for n, v in c["members"].items()} for k, c in sf["interfaces"].items() if c["exported"]},
                    functions={k: dict(params=v["params"], required=v["required"]) for k, v in sf["functions"].items() if v["exported"]})
        out["files"][rel] = keep
    return out


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--reference":
        json.dump(reference_surface(sys.argv[2]), sys.stdout, indent=1, sort_keys=True)
        print()
    else:
        for f in sys.argv[1:]:
            json.dump({f: surface(open(f).read())}, sys.stdout, indent=1, sort_keys=True)
            print()
