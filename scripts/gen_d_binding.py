"""Writes the D declarations (`extern(C)`) of everything include/ptgpu.h declares: constants, structs, functions.

  python3 scripts/gen_d_binding.py            prints the module text
  python3 scripts/gen_d_binding.py --update   rewrites the ```d block of INTEGRATION.md that starts with `module ptgpu;`

tests/test_cabi.py regenerates the text and compares it with that block, so the D text a maintainer copies cannot fall
behind the header.  No D compiler exists in the build environment: the text is checked against the header, not compiled.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALARS = {'int32_t': 'int', 'uint32_t': 'uint', 'uint64_t': 'ulong', 'int64_t': 'long', 'uint8_t': 'ubyte', 'int': 'int',
           'float': 'float', 'double': 'double', 'void': 'void', 'char': 'char'}


def strip_comments(text):
    return re.sub(r'/\*.*?\*/', '', text, flags=re.S)


def d_type(ctype):
    """C parameter / return type (without the name) -> D."""
    t = ctype.strip()
    const = t.startswith('const ')
    if const:
        t = t[6:].strip()
    stars = t.count('*')
    base = t.replace('*', '').strip()
    base = SCALARS.get(base, base)                        # struct names stay
    if stars == 0:
        return base
    if const:
        return 'const(%s)%s' % (base, '*' * stars)
    return base + '*' * stars


def param(p):
    p = p.strip()
    if p == 'void' or not p:
        return None
    m = re.match(r'(.*?)(\w+)\s*\[\s*\w*\s*\]$', p)       # `const float pos[3]` is a pointer parameter
    if m:
        return '%s %s' % (d_type(m.group(1).strip() + '*'), m.group(2))
    m = re.match(r'(.*?)(\w+)$', p)
    name = m.group(2)
    if name in ('in', 'out', 'ref', 'function', 'version'):   # D keywords
        name += '_'
    return '%s %s' % (d_type(m.group(1)), name)


def generate():
    text = strip_comments(open(os.path.join(ROOT, 'include', 'ptgpu.h')).read())
    out = ['module ptgpu;', '// generated from include/ptgpu.h by scripts/gen_d_binding.py', 'extern(C) nothrow @nogc:', '']
    for m in re.finditer(r'#define\s+(PT_\w+)\s+\(?\s*(-?\d+)(u?)\s*(?:<<\s*(\d+))?\s*\)?\s*$', text, flags=re.M):
        name, val, uns, sh = m.groups()
        v = int(val) << int(sh) if sh else int(val)
        out.append('enum %s %s = %d;' % ('uint' if uns else 'int', name, v))
    for m in re.finditer(r'enum\s*\{(.*?)\}\s*;', text, flags=re.S):
        for item in m.group(1).split(','):
            k, v = [x.strip() for x in item.split('=')]
            out.append('enum int %s = %s;' % (k, v))
    out.append('')
    for m in re.finditer(r'typedef\s+struct\s+(\w+)\s+(\w+)\s*;', text):
        out.append('struct %s;' % m.group(2))
    for m in re.finditer(r'typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;', text, flags=re.S):
        fields = []
        for decl in m.group(2).split(';'):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.split(None, 1)
            for n in names.split(','):
                n = n.strip()
                a = re.match(r'(\w+)\s*\[\s*(\w+)\s*\]', n)
                fields.append('%s[%s] %s;' % (SCALARS[ctype], a.group(2), a.group(1)) if a else '%s %s;' % (SCALARS[ctype], n))
        out.append('struct %s { %s }' % (m.group(3), ' '.join(fields)))
    out.append('')
    for m in re.finditer(r'^\s*((?:const\s+)?\w+\s*\**)\s*(ptgpu_\w+)\s*\((.*?)\)\s*;', text, flags=re.M | re.S):
        ret, name, args = m.groups()
        ps = [param(' '.join(p.split())) for p in args.split(',')]
        out.append('%s %s(%s);' % (d_type(ret), name, ', '.join(p for p in ps if p)))
    return '\n'.join(out) + '\n'


def integration_block():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'```d\n(module ptgpu;.*?)```', text, flags=re.S)
    return text, m


if __name__ == '__main__':
    gen = generate()
    if '--update' in sys.argv:
        text, m = integration_block()
        assert m, 'INTEGRATION.md holds no ```d block that starts with `module ptgpu;`'
        open(os.path.join(ROOT, 'INTEGRATION.md'), 'w').write(text[:m.start(1)] + gen + text[m.end(1):])
    else:
        sys.stdout.write(gen)
