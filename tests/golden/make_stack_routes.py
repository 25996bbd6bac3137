"""Extracts the reference's stack-route table into tests/golden/stack_routes.json.  Run in the build container only (reads
/root/reference); the GPU box and the tests read the committed JSON.

Source: air/src/constraints/op_flags/stack_route_tests.rs -- `routes_for_opcode` (per opcode: which of the 16 stack positions keep
their value, take the value from the right, take the value from the left), `aggregate_shifts_for_opcode` (the two scalar shift
flags) and `valid_route_opcodes`; the opcode numbers from core/src/operations/mod.rs (`pub mod opcodes`).  The table is DATA the
reference's test `composite_stack_routes_match_expected_table` checks its `OpFlags` against; tests/test_ref_op_flags.py checks the
hand-ported `OpFlags` against the same table."""
import json, os, re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def opcode_numbers():
    src = open(os.path.join(REF, "core/src/operations/mod.rs")).read()
    body = src[src.index("pub mod opcodes"):]
    out = {}
    for name, val in re.findall(r"pub const (\w+): u8\s*=\s*(0b[01_]+|\d+)\s*;", body):
        out[name] = int(val.replace("_", ""), 0)
    return out


def main():
    opc = opcode_numbers()
    src = open(os.path.join(REF, "air/src/constraints/op_flags/stack_route_tests.rs")).read()
    for name, val in re.findall(r"const (UNUSED_\w+): u8 = (\d+);", src):
        opc[name] = int(val)
    body = src[src.index("fn routes_for_opcode"):src.index("fn aggregate_shifts_for_opcode")]
    body = body[body.index("match opcode {") + len("match opcode {"):body.index("_ => panic!")]
    arm = re.compile(r"((?:\s*(?:opcodes::\w+|UNUSED_\w+)(?:\s+if\s+!?is_loop_end)?\s*\|?)+)\s*=>\s*(\{.*?\}|set\([^;{}]*?\))\s*,", re.S)
    routes = []
    for pats, action in arm.findall(body):
        sets = [dict(flags=f, lo=int(a), hi=int(b)) for f, a, b in re.findall(r"set\(&mut (\w+), (\d+)\.\.(\d+)\)", action)]
        for p in pats.split("|"):
            p = p.strip()
            m = re.match(r"(?:opcodes::)?(\w+)(?:\s+if\s+(!?)is_loop_end)?$", p)
            name, guard = m.group(1), m.group(2)
            when = "always" if "is_loop_end" not in p else ("not_loop_end" if guard == "!" else "loop_end")
            routes.append(dict(name=name, opcode=opc[name], when=when, sets=sets))
    # the unguarded `opcodes::END => ...` arm after the `if !is_loop_end` one is the loop-end case
    seen_end = False
    for r in routes:
        if r["name"] == "END":
            if r["when"] == "always":
                assert seen_end
                r["when"] = "loop_end"
            seen_end = True
    agg = src[src.index("fn aggregate_shifts_for_opcode"):src.index("#[test]")]
    left = agg[agg.index("let left_shift"):agg.index("let right_shift")]
    right = agg[agg.index("let right_shift"):]
    names = lambda t: [opc[n] for n in re.findall(r"(?:opcodes::)?(UNUSED_\w+|(?<=opcodes::)\w+)", t.split(") ||")[0])]   # noqa: E731
    doc = dict(source="air/src/constraints/op_flags/stack_route_tests.rs", routes=routes, left_shift=sorted(set(names(left))),
               left_shift_when_loop_end=[opc["END"]], right_shift=sorted(set(names(right))),
               opcodes={k: v for k, v in opc.items() if not k.startswith("UNUSED")})
    assert len({r["opcode"] for r in routes}) == 64 + 8 + 16 + 8, len({r["opcode"] for r in routes})
    with open(os.path.join(HERE, "stack_routes.json"), "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
    print(len(routes), "arms entries;", len(doc["left_shift"]), "left,", len(doc["right_shift"]), "right")


if __name__ == "__main__":
    main()
