#!/usr/bin/env python3
"""Compiles the drop-in for real (TEST INFRASTRUCTURE): applies INTEGRATION.md's patch to a SCRATCH copy of the
reference's alignment/Blasr.cpp and alignment/MappingParameters.h (made at build time under oracle/_ref/build, never
committed: oracle/_ref is git-ignored), adds oracle/dropin/blasr_hip_dropin.h (the new code, ours), and links
-lblasr_hip into oracle/_ref/blasrmc_gpu.  The edits are the ones a maintainer would make:

  MappingParameters.h   three fields: bool useGPU (false), int gpuDevice (0), int gpuBatch (0)
  Blasr.cpp             #include of the helper before AlignIntervals
                        AlignIntervals (:1716-1724)   SDPAlign + ComputeAlignmentStats -> SDPAlignCandidateGPU under -useGPU
                        AlignIntervals (:1826, :1885) ExtendAlignmentForward / Reverse (-extend) -> ExtendAlignmentGPU under -useGPU
                        RefineAlignments (:2164-2181) the per-candidate loop -> RefineAlignmentsGPU (guided) or RefineAlignmentsKBandGPU
                                                      (-noUseGuidedAlign) under -useGPU
                        MapRead (:2279-2297)          the two MapReadToGenome calls -> MapReadToGenomeGPU under -useGPU
                        MapReads (:3238)              one blasr_hip context per mapping thread, next to MappingBuffers
                        MapReads (:3253, :3470)       -gpuBatch B: reads are pulled B at a time and mapped by ONE library call
                                                      (BlasrHipNextRead); where MapReads calls MapRead it takes the block's
                                                      alignments of that read (BlasrHipCandidatesOfCurrentRead); everything
                                                      after MapRead -- StoreMapQVs, the printers -- is the reference's own
                        main (:4023)                  -useGPU / -gpuDevice / -gpuBatch registered with the command-line parser
                        main (:3947)                  setenv("GPU_MAX_HW_QUEUES", "8", 0) before anything touches HIP
Usage: python oracle/make_dropin.py [reference checkout]   (default /root/reference)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def patch_once(text, old, new, what):
    if text.count(old) != 1:
        raise SystemExit("make_dropin: anchor for %s found %d times" % (what, text.count(old)))
    return text.replace(old, new)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "alignment")
    if not os.path.exists(os.path.join(src, "Blasr.cpp")):
        print("make_dropin: no reference sources at %s: keeping the prebuilt oracle/_ref/blasrmc_gpu (if any)" % ref)
        return
    build = os.path.join(ROOT, "oracle", "_ref", "build")
    os.makedirs(build, exist_ok=True)
    # ---- MappingParameters.h
    mp = open(os.path.join(src, "MappingParameters.h")).read()
    mp = patch_once(mp, "\tbool useGuidedAlign;\n", "\tbool useGuidedAlign;\n\tbool useGPU;\n\tint gpuDevice;\n\tint gpuBatch;\n", "the fields")
    mp = patch_once(mp, "\t\tuseGuidedAlign = true;\n", "\t\tuseGuidedAlign = true;\n\t\tuseGPU = false;\n\t\tgpuDevice = 0;\n\t\tgpuBatch = 0;\n",
                    "the defaults")
    open(os.path.join(build, "MappingParameters.h"), "w").write(mp)
    # MappingIPC.h pulls MappingParameters.h in by a quoted include, i.e. from its own directory first: an unchanged
    # scratch copy next to the patched header makes it find that one
    open(os.path.join(build, "MappingIPC.h"), "w").write(open(os.path.join(src, "MappingIPC.h")).read())
    # ---- Blasr.cpp
    bl = open(os.path.join(src, "Blasr.cpp")).read()
    bl = patch_once(bl, "template<typename T_TargetSequence, typename T_QuerySequence, typename TDBSequence>\nvoid AlignIntervals(",
                    '#include "blasr_hip_dropin.h"\n\ntemplate<typename T_TargetSequence, typename T_QuerySequence, '
                    "typename TDBSequence>\nvoid AlignIntervals(", "the include")
    m = re.search(r"        alignScore = SDPAlign\(alignment->qAlignedSeq, alignment->tAlignedSeq, distScoreFn,.*?"
                  r"params\.sdpMaxAnchorsPerPosition\);\n        ComputeAlignmentStats\(\*alignment, alignment->qAlignedSeq\.seq, "
                  r"alignment->tAlignedSeq\.seq,\n\s*distScoreFn, params\.affineAlign\);\n", bl, flags=re.S)
    if not m:
        raise SystemExit("make_dropin: the SDPAlign call of AlignIntervals was not found")
    bl = (bl[:m.start()] + "        if (params.useGPU) {\n          alignScore = SDPAlignCandidateGPU(*alignment, params, sdpTupleSize);\n"
          "        } else {\n" + m.group(0) + "        }\n" + bl[m.end():])
    bl = patch_once(bl, "  UInt i;\n  for (i = 0; i < alignmentPtrs.size(); i++ ) {\n    RefineAlignment(*bothQueryStrands[0], genome, "
                        "*alignmentPtrs[i], params, mappingBuffers);\n  }\n",
                    "  if (params.useGPU and !params.doGlobalAlignment) {\n    if (params.useGuidedAlign) RefineAlignmentsGPU(*bothQueryStrands[0], "
                    "alignmentPtrs, params);\n    else RefineAlignmentsKBandGPU(*bothQueryStrands[0], alignmentPtrs, params);\n    return;\n  }\n"
                    "  UInt i;\n  for (i = 0; i < alignmentPtrs.size(); i++ ) {\n    RefineAlignment(*bothQueryStrands[0], genome, "
                    "*alignmentPtrs[i], params, mappingBuffers);\n  }\n", "RefineAlignments")
    bl = patch_once(bl, "    if (params.useSuffixArray) {\n      params.anchorParameters.lcpBoundsOutPtr = mapData->lcpBoundsOutPtr;\n",
                    "    if (params.useGPU and params.useSuffixArray) {\n      MapReadToGenomeGPU(genome, sarray, read, params, "
                    "mappingBuffers.matchPosList, mappingBuffers.rcMatchPosList, numKeysMatched, rcNumKeysMatched);\n    }\n"
                    "    else if (params.useSuffixArray) {\n      params.anchorParameters.lcpBoundsOutPtr = mapData->lcpBoundsOutPtr;\n",
                    "MapRead's anchoring")
    bl = patch_once(bl, "          forwardScore = ExtendAlignmentForward(readSuffix, 0,\n",
                    "          forwardScore = params.useGPU ? ExtendAlignmentGPU(readSuffix, 0, genomeSuffix, 0, "
                    "extendedAlignmentForward, params, false) :\n                         ExtendAlignmentForward(readSuffix, 0,\n",
                    "the forward extension")
    bl = patch_once(bl, "          reverseScore = ExtendAlignmentReverse(readPrefix, readPrefix.length-1,\n",
                    "          reverseScore = params.useGPU ? ExtendAlignmentGPU(readPrefix, readPrefix.length-1, genomePrefix, "
                    "genomePrefixLength - 1, extendedAlignmentReverse, params, true) :\n"
                    "                         ExtendAlignmentReverse(readPrefix, readPrefix.length-1,\n", "the reverse extension")
    bl = patch_once(bl, "  MappingBuffers mappingBuffers;\n  while (true) {\n",
                    "  MappingBuffers mappingBuffers;\n  BlasrHipThreadContext hipThreadContext(params.useGPU, params.gpuDevice, params.gpuBatch);\n"
                    "  while (true) {\n", "MapReads")
    # -gpuBatch: reads in blocks (the reader call of MapReads, :3253), MapRead's results from the block (:3470-3496)
    bl = patch_once(bl, "      if (GetNextReadThroughSemaphore(*mapData->reader, params, smrtRead, alignmentContext) == false) {\n",
                    "      if (BlasrHipNextRead(*mapData->reader, params, smrtRead, alignmentContext, genome, sarray, seqdb) == false) {\n",
                    "MapReads' reader call")
    m = re.search(r"        MapRead\(subreadSequence, subreadSequenceRC, \n.*?                mapData           // Some values that are shared\n"
                  r"\s*// across threads\.\n\s*\);\n", bl, flags=re.S)
    if not m:
        raise SystemExit("make_dropin: the MapRead call of MapReads' subread loop was not found")
    bl = (bl[:m.start()] + "        if (params.useGPU and params.gpuBatch > 0) {\n          BlasrHipCandidatesOfCurrentRead(subreadSequence, "
          "subreadSequenceRC, genome, seqdb, params, alignmentPtrs);\n        } else {\n" + m.group(0) + "        }\n" + bl[m.end():])
    bl = patch_once(bl, '  clp.RegisterFlagOption("noUseGuidedAlign", (bool*)&params.useGuidedAlign, "");\n',
                    '  clp.RegisterFlagOption("noUseGuidedAlign", (bool*)&params.useGuidedAlign, "");\n'
                    '  clp.RegisterFlagOption("useGPU", &params.useGPU, "");\n'
                    '  clp.RegisterIntOption("gpuDevice", &params.gpuDevice, "", CommandLineParser::NonNegativeInteger);\n'
                    '  clp.RegisterIntOption("gpuBatch", &params.gpuBatch, "", CommandLineParser::NonNegativeInteger);\n', "main")
    # the HIP runtime reads GPU_MAX_HW_QUEUES once, when it starts: the host asks for eight before anything else (INTEGRATION.md 4')
    bl = patch_once(bl, "int main(int argc, char* argv[]) {\n",
                    'int main(int argc, char* argv[]) {\n  setenv("GPU_MAX_HW_QUEUES", "8", 0);\n', "main's first line")
    open(os.path.join(build, "Blasr_gpu.cpp"), "w").write(bl)
    out = os.path.join(ROOT, "oracle", "_ref", "blasrmc_gpu")
    cmd = ["g++", "-O3", "-std=c++11", "-w", "-I", build, "-I", os.path.join(ROOT, "oracle", "dropin"), "-I", os.path.join(ROOT, "include"),
           "-I", src, "-I", os.path.join(ref, "common"), os.path.join(build, "Blasr_gpu.cpp"), "-L", os.path.join(ROOT, "blasr_amd"),
           "-lblasr_hip", "-Wl,-rpath,$ORIGIN/../../blasr_amd", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib",
           "-lrt", "-lpthread", "-o", out]
    print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
    finally:
        import shutil
        shutil.rmtree(build, ignore_errors=True)  # the scratch copies of the reference's files do not stay around


if __name__ == "__main__":
    main()
