"""State schema of the batched plant stepper, parsed from ``include/npb_fields.h``.

The header is the single source of truth (it also generates the device register
structs); this module only reads it so that Python can
address state columns by name: ``SCHEMA.f64["sg.water_level"][i]`` -> slot.

Slot numbering (must match npb_fields.h): section-major, then instance, then
member order; inside a section all fp64 members precede all int32 members.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")


@dataclass
class Field:
    section: str        # member name of the section ("prim", "sg", ...)
    name: str
    kind: str           # "f64" | "i32"
    count: int          # array length (1 for scalars)
    path: str           # reference attribute path template ("" if none)
    offset: int         # slot offset inside one section instance (per kind)


@dataclass
class Section:
    member: str
    tag: str
    count: int
    fields: List[Field] = field(default_factory=list)
    nf64: int = 0
    ni32: int = 0
    f64_base: int = 0
    i32_base: int = 0
    nout: int = 0      # the last `nout` fp64 members are outputs of the step, stored as float (npb_fields.h)

    @property
    def ncarry(self) -> int:
        return self.nf64 - self.nout

    def ncol(self, storage: str = "f64") -> int:
        """arena columns of one instance: 8-byte columns (two narrow members each) or 4-byte columns (fp32 storage)"""
        narrow = self.nout + self.ni32
        return self.ncarry + ((narrow + 1) // 2 if storage == "f64" else narrow)


class Schema:
    def __init__(self, header_text: str):
        consts = {m.group(1): int(m.group(2))
                  for m in re.finditer(r"#define\s+(NPB_NUM_\w+)\s+(\d+)", header_text)}
        # join continuation lines
        text = header_text.replace("\\\n", " ")
        blocks = {m.group(1): m.group(2)
                  for m in re.finditer(r"#define\s+NPB_(\w+)_FIELDS\(F, A, I\)(.*)", text)}
        nouts = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+NPB_(\w+)_NOUT\s+(\d+)", header_text)}
        sec_line = re.search(r"#define\s+NPB_SECTIONS\(S\)(.*)", text).group(1)
        self.sections: List[Section] = []
        for m in re.finditer(r"S\(\s*(\w+)\s*,\s*(\w+)\s*,\s*(\w+)\s*,\s*(\w+)\s*\)", sec_line):
            member, tag, _stype, cnt = m.groups()
            count = consts[cnt] if cnt in consts else int(cnt)
            sec = Section(member, tag, count)
            body = blocks[tag]
            f64s, i32s = [], []
            for fm in re.finditer(r'\b([FAI])\(\s*(\w+)\s*,\s*(?:(\d+)\s*,\s*)?"([^"]*)"\s*\)', body):
                kind, name, cnt2, path = fm.groups()
                if kind == "I":
                    i32s.append((name, 1, path))
                else:
                    f64s.append((name, int(cnt2) if cnt2 else 1, path))
            off = 0
            for name, c, path in f64s:
                sec.fields.append(Field(member, name, "f64", c, path, off)); off += c
            sec.nf64 = off
            off = 0
            for name, c, path in i32s:
                sec.fields.append(Field(member, name, "i32", c, path, off)); off += c
            sec.ni32 = off
            sec.nout = nouts.get(tag, 0)
            self.sections.append(sec)
        fb = ib = 0
        for sec in self.sections:
            sec.f64_base, sec.i32_base = fb, ib
            fb += sec.count * sec.nf64
            ib += sec.count * sec.ni32
        self.total_f64, self.total_i32 = fb, ib
        self.by_name: Dict[str, Tuple[Section, Field]] = {}
        for sec in self.sections:
            for f in sec.fields:
                self.by_name["%s.%s" % (sec.member, f.name)] = (sec, f)

    def slot(self, name: str, instance: int = 0, k: int = 0) -> Tuple[str, int]:
        """("f64"|"i32", global slot) of field `section.name`, instance, array element k."""
        sec, f = self.by_name[name]
        if not (0 <= instance < sec.count and 0 <= k < f.count):
            raise IndexError(name)
        if f.kind == "f64":
            return "f64", sec.f64_base + instance * sec.nf64 + f.offset + k
        return "i32", sec.i32_base + instance * sec.ni32 + f.offset + k

    def columns(self):
        """Every (kind, slot, label, reference_path) in slot order."""
        out = []
        for sec in self.sections:
            for inst in range(sec.count):
                for f in sec.fields:
                    for k in range(f.count):
                        kind, slot = self.slot("%s.%s" % (sec.member, f.name), inst, k)
                        label = "%s[%d].%s" % (sec.member, inst, f.name) if sec.count > 1 else "%s.%s" % (sec.member, f.name)
                        if f.count > 1:
                            label += "[%d]" % k
                        path = f.path.replace("{i}", str(inst)).replace("{j}", str(inst + 1)).replace("{k}", str(k))
                        out.append((kind, slot, label, path))
        return out

    def state_bytes(self, storage: str = "f64") -> int:
        """Arena bytes per plant: columns of every section instance * column width."""
        width = 8 if storage == "f64" else 4
        return sum(sec.count * sec.ncol(storage) for sec in self.sections) * width

    def n_outputs_step(self) -> int:
        """output members (written by the step, never read by it) of the sections the step kernels own"""
        return sum(sec.count * sec.nout for sec in self.sections if sec.member not in ("mpump", "maint"))

    def is_output(self, name: str) -> bool:
        """True for an fp64 member that the step only writes (stored as float)."""
        sec, f = self.by_name[name]
        return f.kind == "f64" and f.offset >= sec.ncarry


def _load() -> Schema:
    with open(os.path.join(INCLUDE_DIR, "npb_fields.h")) as fh:
        return Schema(fh.read())


def load_params() -> List[Tuple[str, float, str]]:
    with open(os.path.join(INCLUDE_DIR, "npb_params.h")) as fh:
        text = fh.read().replace("\\\n", " ")
    body = re.search(r"#define\s+NPB_PARAM_LIST\(P\)(.*)", text).group(1)
    out = []
    for m in re.finditer(r'P\(\s*(\w+)\s*,\s*([-+0-9.eE]+)\s*,\s*"([^"]*)"\s*\)', body):
        out.append((m.group(1), float(m.group(2)), m.group(3)))
    return out


SCHEMA = _load()
PARAMS = load_params()
