"""The subset of gin-config the reference's launchers use (`gin.parse_config_files_and_bindings(ginc, ginb)`, S3/run.py:283):
`scope.parameter = <python literal>` lines, `#` comments, `include 'file.gin'`, and the same syntax for command-line bindings
(`--ginb "run.max_steps=1000"`).  gin itself is not in this image; the reference's .gin files (configs/*/Backpack.gin) parse
unchanged.  `Bindings.kwargs('run')` gives the keyword arguments gin would inject into `@gin.configurable() def run(...)`."""
from __future__ import annotations

import ast
import os
from typing import Any, Dict, Iterable, Optional


class Bindings(dict):
    """{'run.max_steps': 500000, 'MipNeRF360.opaque_background': True, ...}"""

    def kwargs(self, scope: str) -> Dict[str, Any]:
        p = scope + "."
        return {k[len(p):]: v for k, v in self.items() if k.startswith(p)}

    def get_param(self, name: str, default=None):
        return self.get(name, default)


def _literal(text: str):
    text = text.strip()
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        low = text.lower()
        if low in ("true", "false"):
            return low == "true"
        if low == "none":
            return None
        if text.startswith(("@", "%")):
            return text                     # configurable / macro references are kept verbatim
        raise ValueError(f"gin_lite: cannot parse value {text!r}")


def parse_lines(lines: Iterable[str], out: Optional[Bindings] = None, basedir: str = ".") -> Bindings:
    out = Bindings() if out is None else out
    pending = ""
    for raw in lines:
        line = raw.split("#", 1)[0].rstrip() if not _in_string(raw) else raw.rstrip()
        if not line.strip():
            continue
        line = pending + line
        if line.count("(") > line.count(")") or line.count("[") > line.count("]") or line.endswith("\\"):
            pending = line.rstrip("\\") + " "
            continue
        pending = ""
        s = line.strip()
        if s.startswith("include"):
            parse_file(os.path.join(basedir, _literal(s[len("include"):])), out)
            continue
        if s.startswith("import "):
            continue
        if "=" not in s:
            raise ValueError(f"gin_lite: expected 'name = value', got {s!r}")
        name, value = s.split("=", 1)
        out[name.strip()] = _literal(value)
    return out


def _in_string(line: str) -> bool:
    h = line.find("#")
    if h < 0:
        return False
    before = line[:h]
    return before.count('"') % 2 == 1 or before.count("'") % 2 == 1


def parse_file(path: str, out: Optional[Bindings] = None) -> Bindings:
    with open(path, "r") as f:
        return parse_lines(f.readlines(), out, os.path.dirname(os.path.abspath(path)))


def parse_config_files_and_bindings(config_files, bindings) -> Bindings:
    """Same call shape as gin.parse_config_files_and_bindings: later files / bindings override earlier ones."""
    out = Bindings()
    for f in config_files or []:
        parse_file(f, out)
    parse_lines(list(bindings or []), out)
    return out
