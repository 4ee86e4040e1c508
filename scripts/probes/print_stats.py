import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:44], r['Calls'], "%.1f us" % (float(r['AverageNs']) / 1e3))
