// asm_block.h -- one spelling for an inline-asm block in both builds.
//
// The product build (hipcc, gfx950) expands P2_ASM to a GNU asm statement -- the instruction text, operands and clobbers
// reach the assembler exactly as written.  The test-only emulator build (tests/emu, g++ -DP2HOT_EMU) hands THE SAME
// template string, operand list and clobber list to an instruction interpreter (tests/emu/gcn_asm.h), so the CPU test tier
// executes the instruction streams that ship and checks their hazards and clobbers; it is never part of the product.
//
//   P2_ASM(TEMPLATE, (outputs...), (inputs...), (clobbers...))      P2_ASM_NC(TEMPLATE, (outputs...), (inputs...))
//   an operand is P2_O(name, "constraint", lvalue) / P2_I(name, "constraint", expression); name is [id] or empty (%0, %1, ...)
#pragma once

#define P2_UNPAREN(...) __VA_ARGS__

#ifndef P2HOT_EMU
#define P2_O(name, constraint, var) name constraint(var)
#define P2_I(name, constraint, expr) name constraint(expr)
#define P2_ASM(TEMPLATE, OUTS, INS, CLOBBERS) asm(TEMPLATE : P2_UNPAREN OUTS : P2_UNPAREN INS : P2_UNPAREN CLOBBERS)
#define P2_ASM_NC(TEMPLATE, OUTS, INS) asm(TEMPLATE : P2_UNPAREN OUTS : P2_UNPAREN INS)
#define P2_ASM_INTERPRETED() true
#else
#include "gcn_asm.h"
#define P2_O(name, constraint, var) gcn::out(#name, constraint, var)
#define P2_I(name, constraint, expr) gcn::in(#name, constraint, expr)
#define P2_ASM(TEMPLATE, OUTS, INS, CLOBBERS) gcn::run(TEMPLATE, {P2_UNPAREN OUTS}, {P2_UNPAREN INS}, {P2_UNPAREN CLOBBERS})
#define P2_ASM_NC(TEMPLATE, OUTS, INS) gcn::run(TEMPLATE, {P2_UNPAREN OUTS}, {P2_UNPAREN INS}, {})
#define P2_ASM_INTERPRETED() gcn::interpret()  // off by default: the C fallbacks are much faster
#endif
