#!/usr/bin/env python3
"""Fail-safe for the Rust kits WITHOUT a Rust compiler: every `use` path, associated function, struct field and Cargo
dependency that tools/ref_fixtures/src/main.rs, src/bin/export_dag.rs and src/bin/export_lookup.rs rely on is looked up in a
miden-vm checkout with a small module resolver (file modules, inline modules, `pub use` re-exports incl. aliases and globs,
across workspace crates).  A moved or renamed item fails here instead of on the maintainer's first `cargo build`.

    python tools/ref_fixtures/check_imports.py [/path/to/miden-vm]      (default /root/reference)

What it cannot see: items of crates outside the workspace (Plonky3, wincode: reported as "external"), trait-method
resolution, generics.  Beyond paths it compares, for the calls on the `prove_stark` path (SIGNATURES), the number of arguments at
the kits' call sites and a fragment of each parameter type with the reference's declarations: a hand-made type check of the
calls that matter, not a compiler.
"""
import os
import re
import sys

ITEM_KW = r"(?:fn|struct|enum|trait|type|const|static|union)"


def strip_source(src):
    """Drop comments and string literals (keeps braces balanced for the item splitter)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            depth, i = 1, i + 2
            while i < n and depth:
                if src.startswith("/*", i):
                    depth, i = depth + 1, i + 2
                elif src.startswith("*/", i):
                    depth, i = depth - 1, i + 2
                else:
                    i += 1
        elif c == "r" and re.match(r'r#*"', src[i:i + 8]):
            m = re.match(r'r(#*)"', src[i:])
            end = src.find('"' + m.group(1), i + len(m.group(0)))
            i = n if end < 0 else end + 1 + len(m.group(1))
            out.append('""')
        elif c == '"':
            i += 1
            while i < n and src[i] != '"':
                i += 2 if src[i] == "\\" else 1
            i += 1
            out.append('""')
        elif c == "'" and re.match(r"'(\\.|[^\\'])'", src[i:i + 4]):
            i += len(re.match(r"'(\\.|[^\\'])'", src[i:i + 4]).group(0))
            out.append("' '")
        else:
            out.append(c)
            i += 1
    return "".join(out)


def split_items(body):
    """Top-level items of a module body: text up to a `;` at depth 0, or up to the `}` that closes a depth-0 block."""
    items, depth, start = [], 0, 0
    for i, c in enumerate(body):
        if c in "{([":
            depth += 1
        elif c in "})]":
            depth -= 1
            if depth == 0 and c == "}" and not re.match(r"\s*;", body[i + 1:i + 40]):  # `use a::{b, c};` / `const X: T = T {..};` end at the `;`
                items.append(body[start:i + 1])
                start = i + 1
        elif c == ";" and depth == 0:
            items.append(body[start:i + 1])
            start = i + 1
    return [re.sub(r"^(\s*#!?\[[^\]]*\])+", "", it.strip(), flags=re.S).strip() for it in items if it.strip()]


def parse_use_tree(t):
    """`a::b::{c, d::e as f, g::*}` -> [(path, alias)]; alias '*' = glob."""
    t = t.strip()
    m = re.match(r"^(.*?)(?:::)?\{(.*)\}$", t, re.S)
    if m and t.endswith("}"):
        # find the brace that opens the LAST group at depth 0
        depth, pos = 0, None
        for i, c in enumerate(t):
            if c == "{":
                if depth == 0 and pos is None:
                    pos = i
                depth += 1
            elif c == "}":
                depth -= 1
        prefix = t[:pos].rstrip(":").strip()
        inner, parts, depth, start = t[pos + 1:-1], [], 0, 0
        for i, c in enumerate(inner):
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
            elif c == "," and depth == 0:
                parts.append(inner[start:i])
                start = i + 1
        parts.append(inner[start:])
        out = []
        for part in parts:
            if part.strip():
                for path, alias in parse_use_tree(part):
                    pre = [x.strip() for x in prefix.split("::")] if prefix else []
                    if not path and alias == "self":  # `a::b::{self, ..}` imports a::b itself
                        out.append((pre, pre[-1]))
                    else:
                        out.append((pre + path, alias))
        return out
    m = re.match(r"^(.*?)\s+as\s+(\w+)$", t)
    if m:
        return [([s.strip() for s in m.group(1).split("::")], m.group(2))]
    segs = [s.strip() for s in t.split("::")]
    if segs[-1] == "*":
        return [(segs[:-1], "*")]
    if segs[-1] == "self":
        return [(segs[:-1], segs[-2] if len(segs) > 1 else "self")]
    return [(segs, segs[-1])]


class Module:
    def __init__(self, crate, path, dir_for_children, parent):
        self.crate, self.path, self.child_dir, self.parent = crate, path, dir_for_children, parent
        self.items, self.mods, self.uses = set(), {}, []  # uses: (path, alias, is_pub)

    def load(self, body):
        for it in split_items(body):
            pub = bool(re.match(r"pub\b(?!\s*\()", it))
            core = re.sub(r"^pub(\s*\([^)]*\))?\s*", "", it)
            m = re.match(r"mod\s+(\w+)\s*;", core)
            if m:
                self.mods[m.group(1)] = ("file", pub)
                continue
            m = re.match(r"mod\s+(\w+)\s*\{", core)
            if m:
                sub = Module(self.crate, self.path + [m.group(1)], os.path.join(self.child_dir, m.group(1)), self)
                sub.load(core[core.index("{") + 1:core.rindex("}")])
                self.mods[m.group(1)] = (sub, pub)
                continue
            m = re.match(r"use\s+(.*);$", core, re.S)
            if m:
                for path, alias in parse_use_tree(re.sub(r"\s+", " ", m.group(1))):
                    self.uses.append((path, alias, pub))
                continue
            m = re.match(r"(?:(?:unsafe|async|const|extern\s*\"\"|default)\s+)*" + ITEM_KW + r"\s+(\w+)", core)
            if m and pub:
                self.items.add(m.group(1))
            m = re.match(r"macro_rules!\s*(\w+)", core)
            if m:
                self.crate.macros.add(m.group(1))

    def submodule(self, name):
        ent = self.mods.get(name)
        if ent is None:
            return None
        if ent[0] == "file":
            for cand in (os.path.join(self.child_dir, name + ".rs"), os.path.join(self.child_dir, name, "mod.rs")):
                if os.path.exists(cand):
                    sub = Module(self.crate, self.path + [name], os.path.join(self.child_dir, name), self)
                    sub.load(strip_source(open(cand, errors="ignore").read()))
                    self.mods[name] = (sub, ent[1])
                    return sub
            return None
        return ent[0]


class Crate:
    def __init__(self, name, root_dir):
        self.name, self.dir, self.macros = name, root_dir, set()
        lib = os.path.join(root_dir, "src", "lib.rs")
        self.root = Module(self, [name], os.path.join(root_dir, "src"), None)
        self.ok = os.path.exists(lib)
        if self.ok:
            self.root.load(strip_source(open(lib, errors="ignore").read()))


class Workspace:
    def __init__(self, ref):
        self.ref, self.crates = ref, {}
        for dp, dn, fn in os.walk(ref):
            dn[:] = [d for d in dn if d not in ("target", ".git", "node_modules")]
            if "Cargo.toml" in fn:
                m = re.search(r'^\[package\].*?^name\s*=\s*"([^"]+)"', open(os.path.join(dp, "Cargo.toml")).read(), re.S | re.M)
                if m:
                    self.crates[m.group(1).replace("-", "_")] = (m.group(1), dp)
        self._loaded = {}

    def crate(self, ident):
        if ident not in self._loaded:
            self._loaded[ident] = Crate(ident, self.crates[ident][1]) if ident in self.crates else None
        return self._loaded[ident]

    # -> "ok" | "external" | None (not found)
    def resolve(self, path, ctx=None, seen=None):
        seen = seen or set()
        key = (id(ctx), tuple(path))
        if key in seen:
            return None
        seen.add(key)
        head = path[0]
        if head == "crate" and ctx:
            mod, rest = ctx.crate.root, path[1:]
        elif head == "self" and ctx:
            mod, rest = ctx, path[1:]
        elif head == "super" and ctx and ctx.parent:
            mod, rest = ctx.parent, path[1:]
        elif ctx and (head in ctx.mods or any(a == head for _, a, _ in ctx.uses)):
            mod, rest = ctx, path
        elif head in self.crates:
            c = self.crate(head)
            if not c or not c.ok:
                return None
            mod, rest = c.root, path[1:]
        else:
            return "external"
        return self.walk(mod, rest, seen)

    def walk(self, mod, rest, seen):
        if not rest:
            return "ok"
        name = rest[0]
        if len(rest) == 1 and (name in mod.items or name in mod.crate.macros):
            return "ok"
        sub = mod.submodule(name)
        if sub is not None:
            r = self.walk(sub, rest[1:], seen)
            if r:
                return r
        if name in mod.items:  # an enum variant / associated item below a type: the type itself is what we can check
            return "ok"
        for upath, alias, _pub in mod.uses:
            if alias == name:
                r = self.resolve(upath + rest[1:], mod, seen)
                if r:
                    return r
        external = None
        for upath, alias, _pub in mod.uses:
            if alias == "*":
                r = self.resolve(upath + rest, mod, seen)
                if r == "ok":
                    return r
                external = external or r
        return external  # a glob re-export of a crate outside the workspace may provide the name: cannot be checked here

    def sources(self):
        """{relative path: comment- and string-stripped text} of every .rs file of the checkout (read once)."""
        if not hasattr(self, "_sources"):
            self._sources = {}
            for dp, dn, fn in os.walk(self.ref):
                dn[:] = [d for d in dn if d not in ("target", ".git", "midenhip-fixtures")]
                for f in fn:
                    if f.endswith(".rs"):
                        full = os.path.join(dp, f)
                        self._sources[os.path.relpath(full, self.ref)] = strip_source(open(full, errors="ignore").read())
        return self._sources

    def find_method(self, type_name, method):
        """`fn method` inside some `impl ... Type ... {}` or `trait Type {}` block anywhere in the checkout."""
        pat = re.compile(r"\b(?:impl|trait)\b[^{;]*\b" + re.escape(type_name) + r"\b[^{;]*\{")
        want = re.compile(r"\bfn\s+" + re.escape(method) + r"\b")
        for rel, s in self.sources().items():
            if type_name not in s:
                continue
            for m in pat.finditer(s):
                depth, i = 1, m.end()
                while i < len(s) and depth:
                    depth += s[i] == "{"
                    depth -= s[i] == "}"
                    i += 1
                if want.search(s[m.end():i]):
                    return rel
        return None

    def method_params(self, type_name, method):
        """Parameter list (without the receiver) of `fn method` inside an `impl ... Type` / `trait Type` block: [(name, type text)]."""
        pat = re.compile(r"\b(?:impl|trait)\b[^{;]*\b" + re.escape(type_name) + r"\b[^{;]*\{")
        want = re.compile(r"\bfn\s+" + re.escape(method) + r"\b\s*(?:<[^(]*>)?\s*\(")
        for rel, s in self.sources().items():
            if type_name not in s:
                continue
            for m in pat.finditer(s):
                depth, i = 1, m.end()
                while i < len(s) and depth:
                    depth += s[i] == "{"
                    depth -= s[i] == "}"
                    i += 1
                f = want.search(s[m.end():i])
                if f:
                    body = s[m.end() + f.end():i]
                    return [a for a in split_args(body[:close_paren(body)]) if not re.match(r"^(&\s*)?(mut\s+)?self\b", a)]
        return None

    def find_field(self, struct, field):
        pat = re.compile(r"\bstruct\s+" + re.escape(struct) + r"\b[^{;]*\{")
        want = re.compile(r"\bpub\s+" + re.escape(field) + r"\s*:")
        for rel, s in self.sources().items():
            if struct not in s:
                continue
            for m in pat.finditer(s):
                depth, i = 1, m.end()
                while i < len(s) and depth:
                    depth += s[i] == "{"
                    depth -= s[i] == "}"
                    i += 1
                if want.search(s[m.end():i]):
                    return rel
        return None


def close_paren(s):
    """Index of the `)` closing a list whose `(` has just been consumed."""
    depth = 1
    for i, c in enumerate(s):
        depth += c in "([{"
        depth -= c in ")]}"
        if depth == 0:
            return i
    return len(s)


def split_args(s):
    """Top-level comma-separated pieces (generics' `<,>` count as nesting when balanced)."""
    out, depth, angle, start = [], 0, 0, 0
    for i, c in enumerate(s):
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        elif c == "<":
            angle += 1
        elif c == ">" and angle and s[i - 1] != "-" and s[i - 1] != "=":
            angle -= 1
        elif c == "," and depth == 0 and angle == 0:
            out.append(s[start:i].strip())
            start = i + 1
    out.append(s[start:].strip())
    return [a for a in out if a]


def kit_calls(kit_src, type_name, method):
    """Argument lists of every `Type::method(..)` / `Type::<..>::method(..)` / `.method(..)` call in the kits."""
    pats = [r"\b" + re.escape(type_name) + r"(?:::<[^(]*>)?::" + re.escape(method) + r"\s*\(", r"\." + re.escape(method) + r"\s*\("]
    out = []
    for k, pat in enumerate(pats):
        for m in re.finditer(pat, kit_src):
            body = kit_src[m.end():]
            out.append((k == 0, split_args(body[:close_paren(body)])))
    return out


def kit_uses(src):
    """Every full path imported by a kit source file (its own `use` items, at any depth)."""
    out = []
    for m in re.finditer(r"(?:^|\n)\s*use\s+([^;]+);", strip_source(src)):
        for path, alias in parse_use_tree(re.sub(r"\s+", " ", m.group(1))):
            if alias == "*" or path[0] in ("std", "core", "alloc"):
                continue
            out.append(path)
    return out


# What the kits call beyond their imports: (type or module, associated fn / method), (struct, public field), module-level fns.
METHODS = [
    ("ProverInstance", "new"), ("ProverInstance", "prove"), ("VerifierInstance", "new"), ("VerifierInstance", "verify"),
    ("StarkProof", "from_data"), ("Statement", "new"), ("ProverStatement", "new"), ("ProverStatement", "statement"),
    ("DummyMidenAir", "new"), ("StarkConfig", "challenger"), ("LiftedAir", "num_randomness"), ("LiftedAir", "aux_width"),
    ("LiftedAir", "num_aux_values"), ("LiftedAir", "build_aux_trace"), ("LiftedAir", "eval"), ("LiftedAir", "air_layout"),
    ("MidenMultiAir", "new"), ("MultiAir", "airs"), ("LookupAir", "eval"), ("LookupAir", "num_columns"),
    ("LookupAir", "max_message_width"), ("LookupAir", "num_bus_ids"), ("LookupBuilder", "next_column"), ("LookupColumn", "group"),
    ("LookupGroup", "insert"), ("LookupGroup", "add"), ("LookupGroup", "remove"), ("LookupGroup", "batch"),
    ("LookupMessage", "encode"), ("Challenges", "new"),
]
FIELDS = [("StarkProof", "main_commit"), ("StarkProof", "aux_commit"), ("StarkProof", "quotient_commit"), ("StarkProof", "randomness"),
          ("StarkProof", "alpha"), ("StarkProof", "beta"), ("StarkProof", "z"), ("StarkOutput", "proof"), ("StarkOutput", "digest"),
          ("Challenges", "alpha"), ("Challenges", "beta_powers"), ("Challenges", "bus_prefix")]
CONFIG_ITEMS = ["pcs_params", "poseidon2_config", "rpo_config", "rpx_config", "blake3_256_config", "keccak_config",
                "observe_protocol_params", "RELATION_DIGEST", "LOG_FOLDING_ARITY", "FOLDING_POW_BITS", "DEEP_POW_BITS"]


# Type-level check, one step beyond paths: (type, fn, UFCS?) -> the number of arguments at every call site of the kits must equal the
# number of parameters of the reference's declaration (receiver excluded; a `Trait::method(&obj, ..)` call passes it explicitly),
# and for the prove_stark-path calls the reference's parameter TYPES must contain the given fragments.
SIGNATURES = [
    ("Statement", "new", ["", "Vec<F>", "Vec<F>"]),                         # (multi_air, air_inputs, aux_inputs)   statement.rs
    ("ProverStatement", "new", ["Statement<", "Vec<RowMajorMatrix<F>>"]),   # (statement, traces)
    ("ProverInstance", "new", ["&'a SC", "&'a ProverStatement<", "Option<&'a Preprocessed<"]),
    ("ProverInstance", "prove", ["SC::Challenger"]),
    ("VerifierInstance", "new", ["&'a SC", "&'a Statement<", "Option<"]),
    ("VerifierInstance", "verify", ["&StarkProofData<", "SC::Challenger"]),
    ("StarkProof", "from_data", ["&VerifierInstance<", "&StarkProofData<", "SC::Challenger"]),
    ("DummyMidenAir", "new", ["usize", "usize"]),
    ("MidenMultiAir", "new", []),
]
CONFIG_SIGNATURES = {"pcs_params": 0, "observe_protocol_params": 1, "poseidon2_config": 2, "rpo_config": 2, "rpx_config": 2,
                     "blake3_256_config": 2, "keccak_config": 2}


# Calls into crates OUTSIDE the workspace (p3-air 0.6.2's symbolic builder, wincode): their definitions cannot be seen, but the
# reference's own code calls them the same way (crates/ace-codegen/src/pipeline.rs:71-123, dag/lower.rs:101-246,
# prover/src/lib.rs:347-353) -- each pattern must still occur in the reference's sources.
PRECEDENTS = [
    r"SymbolicAirBuilder::<[^>]*>::new\(", r"\.constraint_layout\(\)", r"\.base_constraints\(\)", r"\.extension_constraints\(\)",
    r"\.degree_multiple\(\)", r"\bbase_indices\b", r"\bext_indices\b", r"\.air_layout\(\)", r"\.periodic_columns\(\)",
    r"BaseEntry::Main\s*\{\s*offset", r"BaseEntry::Preprocessed\s*\{", r"BaseEntry::Public", r"BaseEntry::Periodic",
    r"BaseLeaf::IsFirstRow", r"BaseLeaf::IsLastRow", r"BaseLeaf::IsTransition", r"BaseLeaf::Constant", r"BaseLeaf::Variable",
    r"ExtLeaf::Base\b", r"ExtLeaf::ExtVariable", r"ExtLeaf::ExtConstant", r"ExtEntry::Permutation\s*\{\s*offset", r"ExtEntry::Challenge",
    r"ExtEntry::PermutationValue", r"SymbolicExpression::Leaf", r"SymbolicExpression::Add\s*\{\s*x,\s*y", r"SymbolicExpressionExt::Leaf",
    r"SymbolicExpressionExt::Mul\s*\{\s*x,\s*y", r"num_permutation_challenges", r"num_permutation_values", r"permutation_width",
    r"as_basis_coefficients_slice", r"SerdeCompat<", r"wincode::config::Configuration::default\(\)", r"Felt::new_unchecked\(",
    r"\.as_canonical_u64\(\)", r"RowMajorMatrix::new\(",
]


def check(ref, kit_dir):
    ws = Workspace(ref)
    problems, report = [], []
    # Cargo.toml: every dependency is a workspace dependency that exists
    cargo = open(os.path.join(kit_dir, "Cargo.toml")).read()
    root_cargo = open(os.path.join(ref, "Cargo.toml")).read()
    deps = re.search(r"^\[dependencies\](.*?)^\[", cargo, re.S | re.M).group(1)
    for name in re.findall(r"^([\w-]+)\s*=", deps, re.M):
        in_ws = re.search(r"^" + re.escape(name) + r"\s*=", root_cargo, re.M) is not None
        report.append(("dependency", name, "workspace" if in_ws else "MISSING"))
        if not in_ws:
            problems.append(f"Cargo.toml: `{name}` is not a [workspace.dependencies] entry of the reference")
    for feat_crate, feat in re.findall(r'"([\w-]+)/(\w+)"', cargo):
        d = ws.crates.get(feat_crate.replace("-", "_"))
        if not d or not re.search(r"^" + feat + r"\s*=", open(os.path.join(d[1], "Cargo.toml")).read(), re.M):
            problems.append(f"Cargo.toml: feature `{feat_crate}/{feat}` does not exist")
    m = re.search(r'miden-lifted-stark\s*=\s*\{[^}]*features\s*=\s*\[([^\]]*)\]', cargo)
    for feat in re.findall(r'"(\w+)"', m.group(1)) if m else []:
        if not re.search(r"^" + feat + r"\s*=", open(os.path.join(ws.crates["miden_lifted_stark"][1], "Cargo.toml")).read(), re.M):
            problems.append(f"Cargo.toml: miden-lifted-stark has no feature `{feat}`")
    files = [os.path.join(kit_dir, "src", "main.rs")] + sorted(
        os.path.join(kit_dir, "src", "bin", f) for f in os.listdir(os.path.join(kit_dir, "src", "bin")) if f.endswith(".rs"))
    for f in files:
        for path in kit_uses(open(f).read()):
            r = ws.resolve(path)
            report.append((os.path.basename(f), "::".join(path), r or "NOT FOUND"))
            if r is None:
                problems.append(f"{os.path.basename(f)}: `use {'::'.join(path)}` does not resolve in the reference")
    for item in CONFIG_ITEMS:
        r = ws.resolve(["miden_air", "config", item])
        report.append(("config", item, r or "NOT FOUND"))
        if r != "ok":
            problems.append(f"miden_air::config::{item} not found")
    for t, meth in METHODS:
        where = ws.find_method(t, meth)
        report.append(("method", f"{t}::{meth}", where or "NOT FOUND"))
        if not where:
            problems.append(f"no `fn {meth}` in an impl / trait block of `{t}`")
    corpus = list(ws.sources().values())
    kit_src = "".join(strip_source(open(f).read()) for f in files)
    for pat in PRECEDENTS:
        used = re.search(pat, kit_src) is not None
        found = any(re.search(pat, c) for c in corpus)
        report.append(("precedent", pat, ("in reference" if found else "NOT IN REFERENCE") + ("" if used else " (unused by the kits)")))
        if used and not found:
            problems.append(f"external API pattern /{pat}/ is used by the kits but nowhere in the reference")
    for t, meth, want in SIGNATURES:
        params = ws.method_params(t, meth)
        report.append(("signature", f"{t}::{meth}", "(" + ", ".join(params or []) + ")" if params is not None else "NOT FOUND"))
        if params is None:
            problems.append(f"cannot find the declaration of {t}::{meth}")
            continue
        if len(params) != len(want):
            problems.append(f"{t}::{meth} takes {len(params)} parameter(s) in the reference, the kits assume {len(want)}: {params}")
        for prm, frag in zip(params, want):
            if frag and frag.replace(" ", "") not in prm.replace(" ", ""):
                problems.append(f"{t}::{meth}: parameter `{prm}` does not mention `{frag}`")
        calls = [args for ufcs, args in kit_calls(kit_src, t, meth) if ufcs] or \
                ([args for ufcs, args in kit_calls(kit_src, t, meth)] if meth not in ("new",) else [])
        for args in calls:
            if len(args) != len(want):
                problems.append(f"a kit calls {t}::{meth} with {len(args)} argument(s) {args}, the reference declares {len(params)}")
        report.append(("call sites", f"{t}::{meth}", str(len(calls))))
    cfg_src = ws.sources().get(os.path.join("air", "src", "config.rs"), "")
    for fn_name, n in CONFIG_SIGNATURES.items():
        m = re.search(r"\bpub\s+fn\s+" + fn_name + r"\b\s*(?:<[^(]*>)?\s*\(", cfg_src)
        if not m:
            problems.append(f"air/src/config.rs has no `pub fn {fn_name}`")
            continue
        params = split_args(cfg_src[m.end():][:close_paren(cfg_src[m.end():])])
        report.append(("signature", f"config::{fn_name}", "(" + ", ".join(params) + ")"))
        if len(params) != n:
            problems.append(f"config::{fn_name} takes {len(params)} parameter(s), the kits assume {n}")
        for mm in re.finditer(r"config::" + fn_name + r"\s*\(", kit_src):
            args = split_args(kit_src[mm.end():][:close_paren(kit_src[mm.end():])])
            if len(args) != n:
                problems.append(f"a kit calls config::{fn_name} with {len(args)} argument(s)")
    for s, fld in FIELDS:
        where = ws.find_field(s, fld)
        report.append(("field", f"{s}.{fld}", where or "NOT FOUND"))
        if not where:
            problems.append(f"struct `{s}` has no public field `{fld}`")
    return problems, report


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    problems, report = check(ref, os.path.dirname(os.path.abspath(__file__)))
    for row in report:
        print("%-16s %-90s %s" % row)
    print("\n%d problem(s)" % len(problems))
    for p in problems:
        print("  " + p)
    sys.exit(1 if problems else 0)
