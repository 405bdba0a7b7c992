#!/usr/bin/env python3
"""glsl2cpp.py — mechanical GLSL -> C++ text rewrite of the reference's shader sources (TEST INFRASTRUCTURE).

Reads the shader files from where they lie under /root/reference/shaders (authoring container only) and writes the rewritten
text into a scratch directory given on the command line; nothing is stored in the repository.  Together with glsl_cpu.h
(types, built-ins, images, samplers, ray query) the result compiles with `g++ -std=c++20`, which is how oracle/_ref/
libref_stages.so is made.  No statement of the shaders is changed; the rewrites only translate syntax C++ does not have:

  R1  `#version`, `#extension`, `precision ...;`                      -> removed
  R2  `#ifdef __cplusplus`                                            -> `#ifdef REF_HOST_SIDE` (never defined: the GLSL branch
                                                                          of host_device.h / compress.glsl is the one compiled)
  R3  unsuffixed float literals                                       -> `f` suffix (GLSL literals are single precision)
  R4  parameter qualifiers `in T x` / `out T x` / `inout T x`         -> `T x` / `T& x` / `T& x`
  R5  read swizzles `.xyz` `.xy` `.rgb`                               -> member calls `.xyz()` ...; the two write swizzles
                                                                          `tangent.xyz = e` (a vec3) -> `tangent = e`
  R6  scalar swizzle `attr.tangent.x` (a uint)                        -> `attr.tangent`
  R7  `layout(...)` resource declarations                             -> plain globals (pointer for `T name[]`), buffer_reference
                                                                          blocks -> a struct holding the pointer; local_size -> removed
  R8  `void main()`                                                   -> `void shader_main()`
  R9  functional casts `int(e)` / `uint(e)`                           -> `glsl_int(e)` / `glsl_uint(e)` (the contract's defined
                                                                          float->int conversion, include/rt_detmath.h)
  R10 scalar declarations without initialiser `float x;` (locals, globals, struct members)
                                                                      -> `float x = 0;` (GLSL leaves them undefined; this
                                                                          repository defines undefined values as zero)
  R11 `vecN(... rand(...) ...)`                                       -> `vecN{...}`: GLSL evaluates constructor arguments left to
                                                                          right (spec §6.1.1); C++ only guarantees that for braces
"""
import os
import re
import sys

FILES = ["host_device.h", "globals.glsl", "layouts.glsl", "random.glsl", "common.glsl", "compress.glsl", "reservoir.glsl",
         "pbr_metallicworkflow.glsl", "gltf_material.glsl", "punctual.glsl", "env_sampling.glsl", "sun_and_sky.glsl",
         "shade_state.glsl", "traceray_rq.glsl", "pathtrace.glsl", "denoise_common.glsl",
         "direct_stage.comp", "direct_gen.comp", "direct_reuse.comp", "indirect_stage.comp", "denoise_direct.comp",
         "denoise_indirect.comp", "compose.comp"]

FLOAT_LIT = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])")


def strip_block_comments(s):
    # keep line structure (so compiler diagnostics still point at reference line numbers)
    def repl(m):
        return re.sub(r"[^\n]", " ", m.group(0))
    return re.sub(r"/\*.*?\*/", repl, s, flags=re.S)


def split_line_comment(line):
    i = line.find("//")
    return (line, "") if i < 0 else (line[:i], line[i:])


def brace_rand_ctors(code):
    """R11"""
    out = []
    i = 0
    pat = re.compile(r"\b(vec[234])\(")
    while True:
        m = pat.search(code, i)
        if not m:
            out.append(code[i:])
            break
        start = m.end()  # just after '('
        depth, j = 1, start
        while j < len(code) and depth:
            depth += code[j] == "("
            depth -= code[j] == ")"
            j += 1
        inner = code[start:j - 1]
        if depth == 0 and "rand(" in inner:
            out.append(code[i:m.start()] + m.group(1) + "{" + brace_rand_ctors(inner) + "}")
            i = j
        else:
            out.append(code[i:start])
            i = start
    return "".join(out)


def rewrite(name, text):
    text = strip_block_comments(text)
    lines = []
    for line in text.split("\n"):
        code, com = split_line_comment(line)
        if re.match(r"\s*#\s*(version|extension)\b", code) or re.match(r"\s*precision\s+\w+\s+\w+\s*;", code):   # R1
            lines.append("//" + code + com)
            continue
        code = re.sub(r"#\s*ifdef\s+__cplusplus", "#ifdef REF_HOST_SIDE", code)                                   # R2
        if not re.match(r"\s*#\s*(include|if|ifdef|ifndef|endif|else|elif|undef)\b", code):
            code = FLOAT_LIT.sub(lambda m: m.group(1) + "f", code)                                                 # R3
        lines.append(code + com)
    text = "\n".join(lines)
    # from here on work on the text with line comments blanked out of the way of the regexes
    text = "\n".join(split_line_comment(l)[0] for l in text.split("\n"))

    # R7 resource declarations
    text = re.sub(r"^[ \t]*layout\s*\(\s*buffer_reference[^)]*\)\s*buffer\s+(\w+)\s*\{\s*(\w+)\s+(\w+)\[\]\s*;\s*\}\s*;",
                  r"struct \1 { \2* \3; \1(uint64_t a) : \3((\2*)a) {} };", text, flags=re.M)
    text = re.sub(r"^[ \t]*layout\s*\([^)]*\)\s*buffer\s+\w+\s*\{\s*(\w+)\s+(\w+)\[\]\s*;\s*\}\s*;", r"static \1* \2;", text, flags=re.M)
    text = re.sub(r"^[ \t]*layout\s*\([^)]*\)\s*uniform\s+\w+\s*\{\s*(\w+)\s+(\w+)\s*;\s*\}\s*;", r"static \1 \2;", text, flags=re.M)
    text = re.sub(r"^[ \t]*layout\s*\([^)]*\)\s*uniform\s+(?:readonly\s+)?(\w+)\s+(\w+)\[\]\s*;", r"static \1* \2;", text, flags=re.M)
    text = re.sub(r"^[ \t]*layout\s*\([^)]*\)\s*uniform\s+(?:readonly\s+)?(\w+)\s+(\w+)\s*;", r"static \1 \2;", text, flags=re.M)
    text = re.sub(r"^[ \t]*layout\s*\(\s*local_size[^)]*\)\s*in\s*;", "", text, flags=re.M)
    if re.search(r"^[ \t]*layout\s*\(", text, flags=re.M):
        raise SystemExit(f"{name}: a layout(...) declaration was not understood")

    # R4 parameter qualifiers (only directly after '(' or ',')
    text = re.sub(r"(?<=[(,])(\s*)(?:inout|out)\s+(\w+)\s+", r"\1\2& ", text)
    text = re.sub(r"(?<=[(,])(\s*)in\s+(\w+)\s+", r"\1\2 ", text)
    # R6 / R5
    text = re.sub(r"\.tangent\.x\b", ".tangent", text)
    text, nw = re.subn(r"\b(\w+)\.xyz(\s*)=(?!=)", r"\1\2=", text)
    if name == "shade_state.glsl" and nw != 2:
        raise SystemExit("shade_state.glsl: expected exactly the two `tangent.xyz = ...` write swizzles")
    if name != "shade_state.glsl" and nw:
        raise SystemExit(f"{name}: unexpected write swizzle")
    text = re.sub(r"\.(xyz|xy|rgb)\b(?!\s*\()", r".\1()", text)
    # R8
    text = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", text)
    # R9
    text = re.sub(r"(?<![\w.])int\(", "glsl_int(", text)
    text = re.sub(r"(?<![\w.])uint\(", "glsl_uint(", text)
    # R10
    def init_scalars(m):
        names = [n.strip() for n in m.group(2).split(",")]
        return m.group(1) + " " + ", ".join(n + " = 0" for n in names) + ";"
    text = re.sub(r"(?<![\w&*])(float|int|uint|bool)\s+(\w+(?:\s*,\s*\w+)*)\s*;", init_scalars, text)
    # R11
    text = brace_rand_ctors(text)
    return text


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for f in FILES:
        with open(os.path.join(src, f), "r", encoding="utf-8", errors="replace") as fh:
            t = fh.read()
        with open(os.path.join(dst, f), "w") as fh:
            fh.write(rewrite(f, t))


if __name__ == "__main__":
    main()
