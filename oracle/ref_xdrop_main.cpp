// TEST INFRASTRUCTURE ONLY - never linked into the product.
// Driver that calls the UNMODIFIED reference's gapped x-drop entry points directly
// (XDropFwdFastMem xdropfwdmem.cpp:344, XDropBwdFastMem xdropbwdmem.cpp:23,
// XDropAlignMem xdropalignmem.cpp:217) so that tests/golden/make_golden_xdrop.py can
// record known answers for them.  It is compiled by oracle/build_ref.sh against the
// reference's headers where they lie (-I/root/reference/src) and linked with the
// reference's own objects (minus usearch_main.o); nothing of the reference is copied.
//
// usage: ref_xdrop nt|aa  < cases.txt
//   case line:  F|B|A  X  A  B  [AncLoi AncLoj AncLen]
//   output:     score leni lenj path           (F, B)
//               score loi loj leni lenj path   (A)     path "-" when empty
#include "myutils.h"
#include "alpha.h"
#include "alnparams.h"
#include "xdpmem.h"
#include "objmgr.h"
#include "pathinfo.h"
#include "hsp.h"
#include <string>
#include <iostream>
#include <sstream>

float XDropFwdFastMem(XDPMem &Mem, const byte *A, unsigned LA, const byte *B, unsigned LB,
  const AlnParams &AP, float X, unsigned &Leni, unsigned &Lenj, PathInfo &PI);
float XDropBwdFastMem(XDPMem &Mem, const byte *A, unsigned LA, const byte *B, unsigned LB,
  const AlnParams &AP, float X, unsigned &Leni, unsigned &Lenj, PathInfo &PI);
float XDropAlignMem(XDPMem &Mem, const byte *A, unsigned LA, const byte *B, unsigned LB,
  unsigned AncLoi, unsigned AncLoj, unsigned AncLen, const AlnParams &AP,
  float X, HSPData &HSP, PathInfo &PI);

// the two globals the reference defines next to its own main() (usearch_main.cpp:16-17)
bool g_LowerCaseWarning = false;
bool g_AbortProgress = false;

int main(int argc, char **argv)
	{
	if (argc != 2)
		{
		fprintf(stderr, "usage: ref_xdrop nt|aa < cases\n");
		return 2;
		}
	const bool Nucleo = std::string(argv[1]) == "nt";
	char a0[] = "usearch12", a1[] = "-test", a2[] = "x", a3[] = "-quiet";
	char *Args[] = { a0, a1, a2, a3, 0 };
	MyCmdLine(4, Args);
	InitAlpha();
	AlnParams AP;
	AP.InitFromCmdLine(Nucleo);
	XDPMem Mem;
	ObjMgr &OM = *ObjMgr::CreateObjMgr();
	std::string Line;
	while (std::getline(std::cin, Line))
		{
		if (Line.empty())
			continue;
		std::istringstream ss(Line);
		std::string Mode, sA, sB;
		float X;
		ss >> Mode >> X >> sA >> sB;
		const byte *A = (const byte *) sA.c_str();
		const byte *B = (const byte *) sB.c_str();
		unsigned LA = (unsigned) sA.size(), LB = (unsigned) sB.size();
		PathInfo *PI = OM.GetPathInfo();
		if (Mode == "A")
			{
			unsigned Loi, Loj, Len;
			ss >> Loi >> Loj >> Len;
			HSPData HSP;
			HSP.Loi = HSP.Loj = HSP.Leni = HSP.Lenj = 0;
			HSP.Score = 0;
			float Score = XDropAlignMem(Mem, A, LA, B, LB, Loi, Loj, Len, AP, X, HSP, *PI);
			const char *Path = PI->GetPath();
			printf("%.1f %u %u %u %u %s\n", Score, HSP.Loi, HSP.Loj, HSP.Leni, HSP.Lenj, (Path && *Path) ? Path : "-");
			}
		else
			{
			unsigned Leni = 0, Lenj = 0;
			float Score = Mode == "F" ? XDropFwdFastMem(Mem, A, LA, B, LB, AP, X, Leni, Lenj, *PI)
			  : XDropBwdFastMem(Mem, A, LA, B, LB, AP, X, Leni, Lenj, *PI);
			const char *Path = PI->GetPath();
			printf("%.1f %u %u %s\n", Score, Leni, Lenj, (Path && *Path) ? Path : "-");
			}
		PI->Down();
		}
	return 0;
	}
