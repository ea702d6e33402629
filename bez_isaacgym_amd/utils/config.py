"""Hydra/OmegaConf-compatible config loading without hydra/omegaconf (neither is installed here).

Honours exactly what the reference's configs use (bez_isaacgym/cfg/config.yaml:46-49, train.py:53-58):
  * the `defaults:` list with config groups `task/` and `train/` (`- train: ${task}PPO`),
  * command-line overrides `key=value`, `group=name`, dotted keys `task.env.numEnvs=8`,
  * `${abs.path}` / `${.rel}` / `${..rel}` interpolations,
  * the four resolvers  eq, contains, if, resolve_default.
"""
import copy
import os
import re

import yaml

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cfg")

RESOLVERS = {
    "eq": lambda x, y: str(x).lower() == str(y).lower(),                 # train.py:53
    "contains": lambda x, y: str(x).lower() in str(y).lower(),          # train.py:54
    "if": lambda pred, a, b: a if pred else b,                          # train.py:55
    "resolve_default": lambda default, arg: default if arg == "" else arg,  # train.py:58
}


def _parse_scalar(text):
    """A resolver argument / CLI value written as text -> python value (YAML scalar rules)."""
    t = text.strip()
    if len(t) >= 2 and t[0] == t[-1] and t[0] in "\"'":
        return t[1:-1]
    if t == "":
        return ""
    try:
        v = yaml.safe_load(t)
    except yaml.YAMLError:
        return t
    return v if isinstance(v, (bool, int, float, str, list, dict)) or v is None else t


def _split_top(s, sep=","):
    out, depth, cur, quote = [], 0, "", None
    i = 0
    while i < len(s):
        ch = s[i]
        if quote:
            cur += ch
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
            cur += ch
        elif s.startswith("${", i):
            depth += 1
            cur += "${"
            i += 1
        elif ch == "}" and depth > 0:
            depth -= 1
            cur += ch
        elif ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    out.append(cur)
    return out


class _Resolver:
    def __init__(self, root):
        self.root = root

    def lookup(self, path_keys):
        node = self.root
        for k in path_keys:
            if isinstance(node, list):
                node = node[int(k)]
            else:
                node = node[k]
        return node

    def resolve_ref(self, ref, parent_path):
        """ref: 'a.b.c' absolute or '.a' / '..a' relative to the parent container of the current node."""
        if ref.startswith("."):
            ndots = len(ref) - len(ref.lstrip("."))
            base = list(parent_path[: len(parent_path) - (ndots - 1)]) if ndots > 1 else list(parent_path)
            rest = ref.lstrip(".")
            keys = base + (rest.split(".") if rest else [])
        else:
            keys = ref.split(".")
        val = self.lookup(keys)
        return self.resolve_value(val, keys[:-1])

    def _find_innermost(self, s):
        """Span of the first '${...}' that contains no nested '${'."""
        start = None
        for m in re.finditer(r"\$\{|\}", s):
            if m.group() == "${":
                start = m.start()
            elif start is not None:
                return start, m.end()
        return None

    def resolve_string(self, s, parent_path):
        # resolve innermost expressions first; values substituted into a larger expression are
        # carried through placeholders so that their python type survives
        held = {}
        while True:
            span = self._find_innermost(s)
            if span is None:
                break
            a, b = span
            expr = s[a + 2: b - 1]
            val = self._eval(expr, parent_path, held)
            if a == 0 and b == len(s):
                return val
            key = "\x00%d\x00" % len(held)
            held[key] = val
            s = s[:a] + key + s[b:]
        for k, v in held.items():
            s = s.replace(k, str(v))
        return s

    def _arg(self, text, held):
        t = text.strip()
        if t in held:
            return held[t]
        for k, v in held.items():
            t = t.replace(k, str(v))
        return _parse_scalar(t)

    def _eval(self, expr, parent_path, held):
        m = re.match(r"^\s*([A-Za-z_][A-Za-z0-9_]*)\s*:(.*)$", expr, re.S)
        if m and m.group(1) in RESOLVERS:
            args = [self._arg(a, held) for a in _split_top(m.group(2))]
            return RESOLVERS[m.group(1)](*args)
        ref = expr.strip()
        for k, v in held.items():
            ref = ref.replace(k, str(v))
        return self.resolve_ref(ref, parent_path)

    def resolve_value(self, val, parent_path):
        if isinstance(val, str) and "${" in val:
            return self.resolve_string(val, parent_path)
        return val

    def resolve_tree(self, node, path):
        if isinstance(node, dict):
            return {k: self.resolve_tree(v, path + [k]) for k, v in node.items()}
        if isinstance(node, list):
            return [self.resolve_tree(v, path + [str(i)]) for i, v in enumerate(node)]
        return self.resolve_value(node, path[:-1])


def _set_dotted(d, dotted, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        d = d.setdefault(k, {})
    d[keys[-1]] = value


def load_config(overrides=(), config_name="config", cfg_dir=None, resolve=True):
    """Compose cfg/<config_name>.yaml with its defaults list and `key=value` overrides -> plain dict."""
    cfg_dir = cfg_dir or CFG_DIR
    root = yaml.safe_load(open(os.path.join(cfg_dir, config_name + ".yaml")))
    defaults = root.pop("defaults", [])
    root.pop("hydra", None)
    ov = {}
    for o in overrides:
        if "=" not in o:
            raise ValueError("override must look like key=value: %r" % (o,))
        k, v = o.split("=", 1)
        ov[k.lstrip("+")] = v
    groups = {}
    for item in defaults:
        if isinstance(item, dict):
            for g, name in item.items():
                if "/" in g:
                    continue  # hydra/job_logging etc.
                groups[g] = name
    for g in list(groups):
        if g in ov and os.path.isdir(os.path.join(cfg_dir, g)):
            groups[g] = ov.pop(g)
    for g, name in groups.items():
        if isinstance(name, str) and "${" in name:  # e.g. ${task}PPO -> selected task group name
            name = re.sub(r"\$\{(\w+)\}", lambda m: str(groups[m.group(1)]), name)
            groups[g] = name
        root[g] = yaml.safe_load(open(os.path.join(cfg_dir, g, str(name) + ".yaml")))
    for k, v in ov.items():
        _set_dotted(root, k, _parse_scalar(v))
    if not resolve:
        return root
    return _Resolver(root).resolve_tree(copy.deepcopy(root), [])


def omegaconf_to_dict(d):
    """Kept for drop-in compatibility with utils/reformat.py:33 (configs are already plain dicts here)."""
    return copy.deepcopy(d)


def print_dict(val, nesting=-4, start=True):
    """utils/reformat.py:45 equivalent."""
    if isinstance(val, dict):
        if not start:
            print("")
        nesting += 4
        for k in val:
            print(nesting * " ", end="")
            print(k, end=": ")
            print_dict(val[k], nesting, start=False)
    else:
        print(val)
