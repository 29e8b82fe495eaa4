s|        if (len == 1) {$|        if (len < 0) {|
s|        for (int base = 0; base < len; base += 64) {$|        for (int base = 0; base < -len; base += 64) {|
