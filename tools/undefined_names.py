"""Developer check (no linter in the image): global names a module's functions reference but the module never defines.

    python tools/undefined_names.py file.py [file.py ...]
"""
import ast
import builtins
import symtable
import sys


def module_names(tree):
    names = set(dir(builtins))
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                names.add((a.asname or a.name).split(".")[0])
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            names.add(node.name)
        for t in ast.walk(node) if isinstance(node, (ast.Assign, ast.AugAssign, ast.AnnAssign, ast.For, ast.With, ast.If, ast.Try)) else []:
            if isinstance(t, ast.Name) and isinstance(t.ctx, ast.Store):
                names.add(t.id)
    names.update({"__file__", "__name__", "__doc__"})
    return names


def walk(table, defined, path, out):
    for sym in table.get_symbols():
        if sym.is_referenced() and (sym.is_global() or (table.get_type() == "module" and not sym.is_assigned() and not sym.is_imported())):
            if sym.get_name() not in defined:
                out.add((sym.get_name(), table.get_name()))
    for child in table.get_children():
        walk(child, defined, path, out)


def main():
    bad = False
    for path in sys.argv[1:]:
        src = open(path).read()
        defined = module_names(ast.parse(src))
        out = set()
        walk(symtable.symtable(src, path, "exec"), defined, path, out)
        for name, where in sorted(out):
            print(f"{path}: undefined global {name!r} (used in {where})")
            bad = True
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
