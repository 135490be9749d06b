"""tests/cpp/pcl_mock/pcl_mock.hpp is what include/pclhip/pcl_plugin.hpp has been compiled against (real PCL needs Eigen,
Boost and FLANN, none installed): 1,100 lines of TRANSCRIBED signatures.  This test ties the transcription to the
headers it cites -- a drifted signature in the mock would let an `override` of the binding compile here and fail against
real PCL.  For every `virtual` / `override` member function the mock declares, the header named by the comment in front
of its class (path relative to the PCL tree) must declare a member function with the same name, the same parameter types,
the same cv-qualifier and the same return type -- and declare it virtual (or override) as well --, after whitespace, parameter names, default arguments and `pcl::`
qualifications are normalised away.  Needs /root/reference (skipped where it does not exist, e.g. on the GPU box).
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
MOCK = os.path.join(ROOT, "tests", "cpp", "pcl_mock", "pcl_mock.hpp")

BUILTIN = {"int", "float", "double", "bool", "char", "long", "short", "unsigned", "signed", "void", "const", "auto"}
DROP_SPECIFIERS = {"virtual", "inline", "static", "explicit", "PCL_EXPORTS", "typename", "constexpr"}
# headers a class's members may also come from (the mock flattens some of PCL's inheritance into the class it cites)
ALSO = {
    "registration/include/pcl/registration/icp.h": ["registration/include/pcl/registration/registration.h"],
    "search/include/pcl/search/kdtree.h": ["search/include/pcl/search/search.h"],
    "features/include/pcl/features/normal_3d.h": ["features/include/pcl/features/feature.h"],
    "filters/include/pcl/filters/voxel_grid.h": ["filters/include/pcl/filters/filter.h"],
    "filters/include/pcl/filters/filter.h": ["common/include/pcl/pcl_base.h"],
    "registration/include/pcl/registration/correspondence_rejection.h": [
        "registration/include/pcl/registration/correspondence_rejection_distance.h",
        "registration/include/pcl/registration/correspondence_rejection_median_distance.h",
        "registration/include/pcl/registration/correspondence_rejection_one_to_one.h",
        "registration/include/pcl/registration/correspondence_rejection_trimmed.h"],
}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def tokens(s):
    return re.findall(r"::|&&|[A-Za-z_]\w*|\d+|[^\sA-Za-z_\d]", s)


def split_params(s):
    out, depth, cur = [], 0, []
    for t in tokens(s):
        if t in "<([{":
            depth += 1
        elif t in ">)]}":
            depth -= 1
        if t == "," and depth == 0:
            out.append(cur)
            cur = []
        else:
            cur.append(t)
    if cur:
        out.append(cur)
    return out


def norm_type(toks):
    """Tokens of one parameter (or of a return type) -> canonical type string: default argument, parameter name,
    `pcl::` / `typename` and the position of `const` removed."""
    depth = 0
    for i, t in enumerate(toks):  # default argument
        if t in "<([{":
            depth += 1
        elif t in ">)]}":
            depth -= 1
        elif t == "=" and depth == 0:
            toks = toks[:i]
            break
    toks = [t for t in toks if t not in DROP_SPECIFIERS]
    if len(toks) >= 2 and re.match(r"[A-Za-z_]\w*$", toks[-1]) and toks[-1] not in BUILTIN and \
            (toks[-2] in ("&", "*", ">", "&&") or (re.match(r"[A-Za-z_]\w*$", toks[-2]) and toks[-2] != "const" or toks[-2] in BUILTIN - {"const"})):
        toks = toks[:-1]  # the parameter's name
    out, i = [], 0
    while i < len(toks):  # pcl:: qualification
        if toks[i] == "pcl" and i + 1 < len(toks) and toks[i + 1] == "::":
            i += 2
            continue
        out.append(toks[i])
        i += 1
    consts = out.count("const")
    out = [t for t in out if t != "const"]
    if "&" not in out and "*" not in out:
        consts = 0  # top-level const of a by-value parameter is not part of the signature
    return ("const " * min(consts, 1)) + "".join(out)


def norm_return(toks):
    """Return types are spelled through different aliases of the same class (`Ptr`, `Base::Ptr`,
    `typename CorrespondenceEstimationBase<S, T, Scalar>::Ptr`): the last component of the qualified name counts."""
    s = norm_type(toks)
    depth, last = 0, 0
    for i, ch in enumerate(s):
        depth += ch == "<"
        depth -= ch == ">"
        if depth == 0 and s[i:i + 2] == "::":
            last = i + 2
    return s[last:] if not s.startswith("const ") else "const " + s[last:] if last else s


def declarations(text, only_virtual):
    """(name, return type, [param types], const?, line) of the member functions declared in `text`."""
    text = strip_comments(text.replace("\\\n", " "))  # (macro bodies: line continuations)
    found = []
    for m in re.finditer(r"([A-Za-z_]\w*)\s*\(", text):
        name = m.group(1)
        if name in ("if", "for", "while", "switch", "return", "sizeof", "static_assert", "defined", "decltype", "catch",
                    "PCL_DEPRECATED", "PCL_ERROR", "PCL_WARN", "PCL_DEBUG", "PCL_MAKE_ALIGNED_OPERATOR_NEW", "operator"):
            continue
        # balanced parameter list
        i, depth = m.end(), 1
        while i < len(text) and depth:
            depth += text[i] == "("
            depth -= text[i] == ")"
            i += 1
        if depth:
            continue
        params = text[m.end():i - 1]
        tail = re.match(r"\s*(const)?\s*(noexcept)?\s*(override|final)?\s*(=\s*0)?\s*[;{]", text[i:i + 80])
        if not tail:
            continue
        # what stands in front of the name, back to the previous statement / access specifier
        j = m.start()
        k = j
        while k > 0 and text[k - 1] not in ";{}":
            k -= 1
        head = text[k:j]
        head = re.sub(r"^.*\b(public|protected|private)\s*:", " ", head, flags=re.S)
        head = re.sub(r"template\s*<[^;{}]*?>\s*(?=[A-Za-z_:])", " ", head, count=1, flags=re.S) if "template" in head else head
        head_toks = tokens(head)
        if not head_toks or head_toks[-1] in ("::", ".", "->", ",", "(", "=", "return", "new", ":", "?", "!", "+", "-", "<"):
            continue  # a call or a constructor initialiser, not a declaration
        is_virtual = "virtual" in head_toks or bool(tail.group(3))
        if only_virtual and not is_virtual:
            continue
        ret = norm_return([t for t in head_toks if t != "virtual"])
        if ret == "":
            continue  # constructor / destructor
        ptypes = [norm_type(p) for p in split_params(params)]
        ptypes = [p for p in ptypes if p not in ("", "void")]
        found.append((name, ret, ptypes, bool(tail.group(1)), text.count("\n", 0, m.start()) + 1, is_virtual))
    return found


def mock_sections():
    """[(cited header, text of the class section)]: a `// <path>.h:lines` comment at column 0 opens a section."""
    lines = open(MOCK).read().split("\n")
    sections, cur_hdr, cur = [], None, []
    for ln in lines:
        m = re.match(r"// ((?:common|search|kdtree|registration|features|filters)/include/pcl/[\w/]+\.h)\b", ln)
        if m:
            if cur_hdr:
                sections.append((cur_hdr, "\n".join(cur)))
            cur_hdr, cur = m.group(1), []
        cur.append(ln)
    if cur_hdr:
        sections.append((cur_hdr, "\n".join(cur)))
    return sections


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the PCL tree at /root/reference")
def test_every_virtual_of_the_mock_is_declared_that_way_by_the_header_it_cites():
    checked, problems = 0, []
    for hdr, text in mock_sections():
        paths = [hdr] + ALSO.get(hdr, [])
        ref_decls = []
        for p in paths:
            full = os.path.join(REF, p)
            assert os.path.exists(full), "the mock cites a header the reference does not have: " + p
            ref_decls += declarations(open(full, errors="replace").read(), only_virtual=False)
        by_name = {}
        for d in ref_decls:
            by_name.setdefault(d[0], []).append(d)
        for name, ret, ptypes, is_const, line, _ in declarations(text, only_virtual=True):
            checked += 1
            cands = by_name.get(name, [])
            if not any(c[2] == ptypes and c[3] == is_const and c[1] == ret and c[5] for c in cands):  # c[5]: virtual THERE too
                problems.append("%s: %s %s(%s)%s -- the header has: %s" % (
                    hdr, ret, name, ", ".join(ptypes), " const" if is_const else "",
                    "; ".join("%s%s (%s)%s" % ("virtual " if c[5] else "NON-VIRTUAL ", c[1], ", ".join(c[2]), " const" if c[3] else "") for c in cands) or "no such member"))
    assert checked >= 60, "the parser lost the mock's declarations (%d found)" % checked
    assert not problems, "\n".join(problems)


def test_the_normaliser_itself():
    t = lambda s: norm_type(tokens(s))
    assert t("const PointT &point") == t("const PointT& p") == t("const PointT&") == "const PointT&"
    assert t("unsigned int max_nn = 0") == t("unsigned int") == "unsignedint"
    assert t("pcl::Indices &k_indices") == t("Indices&") == "Indices&"
    assert t("std::vector<std::vector<float> > &k_sqr_distances") == t("std::vector<std::vector<float>>&")
    assert t("double max_distance = std::numeric_limits<double>::max ()") == t("double") == "double"
    assert t("const PointCloudConstPtr& cloud") != t("PointCloudConstPtr& cloud")
    d = declarations("struct A { virtual int\n f (const X &x, int k = 3) const = 0;\n void g() override {} };", True)
    assert [(x[0], x[1], x[2], x[3]) for x in d] == [("f", "int", ["const X&", "int"], True), ("g", "void", [], False)]


# The binding also leans on PROTECTED DATA members and non-virtual members of PCL's classes (this->tree_,
# this->force_no_recompute_, this->corr_dist_threshold_, ...): every name it reaches through `this->` or `Base::` must be
# an identifier of the headers its base classes live in, so that a renamed member of the mock cannot hide a break.
PLUGIN = os.path.join(ROOT, "include", "pclhip", "pcl_plugin.hpp")
BASE_HEADERS = [
    "common/include/pcl/pcl_base.h", "search/include/pcl/search/search.h", "search/include/pcl/search/kdtree.h",
    "registration/include/pcl/registration/registration.h", "registration/include/pcl/registration/icp.h",
    "registration/include/pcl/registration/correspondence_estimation.h",
    "registration/include/pcl/registration/transformation_estimation.h",
    "registration/include/pcl/registration/default_convergence_criteria.h",
    "registration/include/pcl/registration/convergence_criteria.h",
    "features/include/pcl/features/feature.h", "features/include/pcl/features/normal_3d.h",
    "filters/include/pcl/filters/filter.h", "filters/include/pcl/filters/voxel_grid.h",
]


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the PCL tree at /root/reference")
def test_every_base_class_member_the_binding_touches_exists_in_pcl():
    plugin = strip_comments(open(PLUGIN).read())
    used = set(re.findall(r"this->\s*([A-Za-z_]\w*)", plugin)) | set(re.findall(r"\bBase::([A-Za-z_]\w*)", plugin))
    own = set(re.findall(r"^\s*(?:[\w:<>,\s\*&]+?)\s+([a-z]\w*_)\s*(?:=[^;]*)?;", plugin, flags=re.M))  # the binding's own members
    ref_ids = set()
    for h in BASE_HEADERS:
        ref_ids |= set(re.findall(r"[A-Za-z_]\w*", strip_comments(open(os.path.join(REF, h), errors="replace").read())))
    missing = sorted(n for n in used if n not in ref_ids and n not in own and n not in ("estimatorKind", "FOREIGN"))
    assert len(used) >= 40, sorted(used)
    assert not missing, "pcl_plugin.hpp reaches for members PCL's headers do not have: %s" % missing
