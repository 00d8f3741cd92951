"""Rewrites the Rust block of INTEGRATION.md section 2 from bindings/rust/raytracer-mi355x-sys/src/lib.rs, the one
place the -sys declarations are maintained (tests/test_abi_symbols.py compares the two token for token).
Usage: python tools/gen_integration.py [--check]"""
import os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- BEGIN lib.rs -->", "<!-- END lib.rs -->"


def rust_block():
    src = open(os.path.join(ROOT, "bindings", "rust", "raytracer-mi355x-sys", "src", "lib.rs")).read()
    body = src[src.index("#![allow"):]  # the crate-level doc comment stays in the crate
    return "```rust\n" + body.rstrip() + "\n```"


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    md = open(path).read()
    a, b = md.index(BEGIN) + len(BEGIN), md.index(END)
    new = md[:a] + "\n" + rust_block() + "\n" + md[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == md else 1)
    open(path, "w").write(new)


if __name__ == "__main__":
    main()
