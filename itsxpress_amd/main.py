"""create_runtime_hmm: the profile selection step of the path (mirror of itsxpress/main.py:176-231).

Same name, arguments, return value and file side effect as the reference function: it
writes `<tempdir>/runtime_selected.hmm` holding, in source order, every model whose NAME
starts with one of the region's prefixes.  File order matters downstream: ItsPosition keeps
the FIRST strictly-highest score, so equal scores are won by the earlier profile.
"""
import logging
import os

REGION_PREFIXES = {"ITS2": ("3_", "4_"), "ITS1": ("1_", "2_"), "ALL": ("1_", "4_")}


def select_profile_blocks(paths, region):
    """Yield the text of every selected model block from the given HMMER3/f files."""
    prefixes = REGION_PREFIXES.get(region, ("1_", "2_", "3_", "4_"))
    for path in paths:
        if not os.path.exists(path):
            # the reference skips a missing taxon file silently (main.py:214-215), which turns `--taxa Fungi` into an EMPTY
            # runtime_selected.hmm and a run that trims nothing; same result here, but not silently
            logging.warning("create_runtime_hmm: profile file %s not found: no profiles selected from it "
                            "(set ITSXPRESS_DB_DIR to a directory that holds ITSx_db/HMMs)", path)
            continue
        with open(path, "r") as fh:
            block, keep = [], False
            for line in fh:
                block.append(line)
                if line.startswith("NAME  ") and line[6:].strip().startswith(prefixes):
                    keep = True
                if line.strip() == "//":
                    if keep:
                        yield "".join(block)
                    block, keep = [], False


def create_runtime_hmm(taxa: str, region: str, tempdir: str) -> str:
    from .definitions import ROOT_DIR, taxa_dict

    hmm_dir = os.path.join(ROOT_DIR, "ITSx_db", "HMMs")
    if taxa in ("All", "all.hmm"):
        files = [os.path.join(hmm_dir, f) for t, f in taxa_dict.items() if t != "All" and f != "all.hmm"]
    else:
        files = [os.path.join(hmm_dir, taxa_dict.get(taxa, taxa))]
    target = os.path.join(tempdir, "runtime_selected.hmm")
    with open(target, "w") as out:
        for text in select_profile_blocks(files, region):
            out.write(text)
    return target
