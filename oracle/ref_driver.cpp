// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY: a command-line driver around the REFERENCE's own C++ variant
// (tools/montecarlo_cpp/Montecarlo.{h,cpp}), compiled from the sources where they lie under /root/reference by
// oracle/Makefile (`make -C oracle ref`) into oracle/_ref/ref_mc.  No reference source is copied into this repo.
//
// The reference's C++ deals with std::shuffle seeded from std::random_device every iteration
// (Montecarlo.cpp:297), i.e. UNIFORMLY and not reproducibly, so it can only be a statistical cross-check: it
// pins the "uniform" dealing law of this repo, and it is timed as cpu_baseline kind "reference".
//
//   ref_mc equity <card1> <card2> <players> <iterations> [table cards...]   -> "equity seconds"
//   ref_mc best                      (stdin: one showdown per line: hands of 7 cards separated by '|')
//                                    -> one line per showdown: 1 if the FIRST hand is best (ties included) else 0
#include <chrono>
#include <iostream>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "Montecarlo.h"

int main(int argc, char **argv) {
    if (argc >= 6 && std::string(argv[1]) == "equity") {
        std::set<std::string> mine{argv[2], argv[3]}, table;
        int players = std::stoi(argv[4]), iters = std::stoi(argv[5]);
        for (int i = 6; i < argc; i++) table.insert(argv[i]);
        auto t0 = std::chrono::steady_clock::now();
        double eq = montecarlo(mine, table, players, iters);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout.precision(9);
        std::cout << eq << " " << dt << std::endl;
        return 0;
    }
    if (argc >= 2 && std::string(argv[1]) == "best") {
        std::string line;
        while (std::getline(std::cin, line)) {
            std::vector<CardsWithTableCombined> hands;
            std::stringstream ss(line);
            std::string part;
            while (std::getline(ss, part, '|')) {
                std::stringstream hs(part);
                CardsWithTableCombined h;
                std::string c;
                while (hs >> c) h.insert(c);
                if (!h.empty()) hands.push_back(h);
            }
            if (hands.empty()) continue;
            std::cout << (eval_best_hand(hands) ? 1 : 0) << "\n";
        }
        return 0;
    }
    std::cerr << "usage: ref_mc equity c1 c2 players iterations [table...] | ref_mc best < showdowns" << std::endl;
    return 2;
}
